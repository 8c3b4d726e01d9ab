## post_hip.R -- the rows either side of the test with the MI355X backend behind them (SURVEY.md §8 f3, f4).
##
##   .getRegionUniverseHip(chicdiff.settings, suffix = "") chicdiff.R:369-426 (window mode)
##   .hipRegionDistances(block)                            chicdiff.R:1965, :1979: IHWcorrection()'s covariate for a device block
##   .hipApplyIHWweights(out, distLookup)                  chicdiff.R:2036-2049, the part of IHWcorrection() after
##                                                         ihw() has been trained on the control set
## The reference's getRegionUniverse() / IHWcorrection() call these when chicdiff.settings[["backend"]] == "hip"
## (r/patches/chicdiff_hip.patch: one line in the former, two changed lines and a four-line bracket in the latter).
## NOT run in this repository (no R here); the tested twins are chicdiff_amd/post.py:getRegionUniverse and
## applyIHWweights (tests/test_gpu_parity.py: region universe against a literal restatement of chicdiff.R:353-401 on
## the reference's chr19 fragments; IHW application against the reference's own result table, 24 863 rows).

.getRegionUniverseHip <- function(chicdiff.settings, suffix = "") {

  RUexpand <- chicdiff.settings[["RUexpand"]]
  rmapfile <- chicdiff.settings[["rmapfile"]]
  chicagoData <- chicdiff.settings[["chicagoData"]]
  saveAux <- chicdiff.settings[["saveAuxData"]]
  outprefix <- chicdiff.settings[["outprefix"]]
  device <- .hipDeviceIndex(chicdiff.settings)   # the new key `hipDevice`; `device` stays the reference's plot device

  ## reading and filtering the peak matrix stays the reference's R (chicdiff.R:230-277)
  x <- readAndFilterPeakMatrix(peakFiles = chicdiff.settings[["peakfiles"]], score = chicdiff.settings[["score"]],
                               targetColumns = chicdiff.settings[["targetColumns"]], chicagoData = chicagoData,
                               conditions = names(chicagoData), outprefix = outprefix)

  ## chromosome of every fragment ID on the map, -1 for an ID the map does not hold: what the two rmap joins and the
  ## `otherEndID <= maxfrag` filter of chicdiff.R:382-399 look up
  rmap <- .hipReadRmap(rmapfile)
  maxfrag <- max(rmap$ID)
  chr_of <- rep(-1L, maxfrag + 1L)                      # entry ID + 1 (IDs start at 1; entry 1 = ID 0 is never on a map)
  chr_of[rmap$ID + 1L] <- as.integer(factor(rmap$chr)) - 1L

  ## .expandAvoidBait() for every call, clipped to the map and to the bait's chromosome, rows in (regionID, otherEndID)
  ## order; regionID <- 1:nrow(x) as chicdiff.R:391
  ru <- .Call("chicdiff_hip_region_universe", .hipContext(device), as.integer(x$baitID), as.integer(x$oeID),
              as.integer(RUexpand), chr_of, PACKAGE = "chicdiffhip")
  RU.DT <- data.table::data.table(baitID = ru$baitID, regionID = ru$regionID, otherEndID = ru$otherEndID)
  data.table::setkey(RU.DT, otherEndID)                 # the reference's two setkey() calls (chicdiff.R:389, :393):
  data.table::setkey(RU.DT, baitID)                     # its row order, which getFullRegionData relies on

  if (saveAux == TRUE) saveRDS(RU.DT, paste0(outprefix, "_RegionUniverse", suffix, ".Rds"))
  RU.DT
}

## IHWcorrection(), chicdiff.R:2036-2049: group cut of log|avDist| by the control set's distance bins, weight look-up,
## renormalisation, weighted p-values and their BH adjustment -- columns added to `out` in place, as the reference does.
## `distLookup` as built at chicdiff.R:2012-2031 (group, minLogDist with [1] <- 0, maxLogDist with [n] <- Inf, avWeights).
.hipApplyIHWweights <- function(out, distLookup, device = 0L) {
  out[, avgLogDist := log(abs(avDist))]
  breaks <- (c(distLookup$minLogDist, Inf) + c(0, distLookup$maxLogDist)) / 2
  w <- .Call("chicdiff_hip_ihw_apply", .hipContext(device), as.double(out$avDist), as.double(out$pvalue), as.double(breaks),
             as.double(distLookup$avWeights), PACKAGE = "chicdiffhip")
  ## (the reference merges on `group`, which re-sorts `out` by group; the values per row are the same)
  out[, group := w$group]
  out[, avWeights := distLookup$avWeights[w$group]]
  out[, weight := w$weight]
  out[, weighted_pvalue := w$weighted_pvalue]
  out[, weighted_padj := w$weighted_padj]
  data.table::setcolorder(out, c("group", setdiff(names(out), "group")))   # merge(out, distLookup, by = "group") puts the key first
  data.table::setkey(out, group)
  out
}

## IHWcorrection() (chicdiff.R:1956-2065) stays the reference's own function: r/patches/chicdiff_hip.patch lets the device path in
## at two points only -- the per-region covariate (chicdiff.R:1965, :1979) and the application block (:2038-2049) -- so ihw()
## training, the distance look-up, the diagnostic plots and saveRDS are the reference's statements, not restated here.

## is this the device-resident block getFullRegionData() returns with backend = "hip"?
.isHipRegionData <- function(x) inherits(x, "chicdiffHipRegionData")

## RU.recast[, list(avDist = mean(distSign)), by = "regionID"] for a block: the per-region covariate was formed on the device when the
## block was built (chicdiff_hip_region_avdist); rows are in regionID order, which is how the reference assigns it (by position)
.hipRegionDistances <- function(block) {
  n <- block$n
  data.table::data.table(regionID = seq_len(n), avDist = .Call("chicdiff_hip_download", block$avDist, PACKAGE = "chicdiffhip"))
}
