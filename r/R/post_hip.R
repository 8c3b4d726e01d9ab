## post_hip.R -- the rows either side of the test with the MI355X backend behind them (SURVEY.md §8 f3, f4).
##
##   getRegionUniverse(chicdiff.settings, suffix = "")     chicdiff.R:369-426 (window mode)
##   .hipApplyIHWweights(out, distLookup)                  chicdiff.R:2036-2049, the part of IHWcorrection() after
##                                                         ihw() has been trained on the control set
## Same conventions as r/R/DESeq2Wrap_hip.R: chicdiff.settings[["backend"]] == "hip" selects the device path, anything
## else the reference's own function (kept by the maintainer as .getRegionUniverseReference, INTEGRATION.md).
## NOT run in this repository (no R here); the tested twins are chicdiff_amd/post.py:getRegionUniverse and
## applyIHWweights (tests/test_gpu_parity.py: region universe against a literal restatement of chicdiff.R:353-401 on
## the reference's chr19 fragments; IHW application against the reference's own result table, 24 863 rows).

getRegionUniverse <- function(chicdiff.settings, suffix = "") {

  if (!identical(chicdiff.settings[["backend"]], "hip"))
    return(.getRegionUniverseReference(chicdiff.settings, suffix = suffix))

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
  rmap <- data.table::fread(rmapfile)
  data.table::setnames(rmap, c("chr", "start", "end", "ID"))
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

## IHWcorrection(), chicdiff.R:1956-2065, for the device path.  Same signature as the reference.  FullRegionData /
## FullControlRegionData may be the "chicdiffHipRegionData" blocks of getFullRegionData(): the per-region covariate
## avDist = mean(distSign) (chicdiff.R:1965-1967, :1980-1982) then comes from the device (chicdiff_hip_region_avdist)
## instead of a group-by over the long table, and the application side (:2036-2049) runs on the device.  ihw() training
## (:1994), the distance look-up (:2004-2031) and the plots (:1999-2002, :2053-2060) are the reference's own statements.
## With long tables and any other backend the reference's function runs (kept as .IHWcorrectionReference).
IHWcorrection <- function(chicdiff.settings, DESeqOut, FullRegionData, DESeqOutControl, FullControlRegionData,
                          countput, DiagPlot = TRUE, diffbaitPlot = TRUE, suffix = "") {

  hip <- inherits(FullRegionData, "chicdiffHipRegionData") && inherits(FullControlRegionData, "chicdiffHipRegionData")
  if (!hip)
    return(.IHWcorrectionReference(chicdiff.settings, DESeqOut, FullRegionData, DESeqOutControl, FullControlRegionData,
                                   countput, DiagPlot = DiagPlot, diffbaitPlot = diffbaitPlot, suffix = suffix))

  baitmapfile <- chicdiff.settings[["baitmapfile"]]
  device <- chicdiff.settings[["device"]]          # the PLOT device, as in the reference (chicdiff.R:1960, :2058)
  outprefix <- chicdiff.settings[["outprefix"]]
  gpu <- .hipDeviceIndex(chicdiff.settings)

  out <- data.table::copy(DESeqOut)
  out$avDist <- .Call("chicdiff_hip_download", FullRegionData$avDist, PACKAGE = "chicdiffhip")   # by position = regionID order
  out$uniform <- runif(nrow(out))
  out$shuff <- sample(out$pvalue)
  message("Comparison against p-vals for out")
  data.table::setDT(out)

  out.control <- data.table::copy(DESeqOutControl)
  out.control$avDist <- .Call("chicdiff_hip_download", FullControlRegionData$avDist, PACKAGE = "chicdiffhip")
  out.control$uniform <- runif(nrow(out.control))
  out.control$shuff <- sample(out.control$pvalue)
  message("Comparison against p-vals for outcontrol")

  ## Train weights on the control sample (IHW stays R)
  ihwRes <- IHW::ihw(pvalue ~ abs(avDist), data = as.data.frame(out.control), alpha = 0.05)
  message("Trained weights on the control sample")
  if (DiagPlot == TRUE) {
    plot(ihwRes)
    ggplot2::ggsave(paste0(outprefix, "_IHWweightPlot.png"), device = "png", path = "./")
    plot(ihwRes, what = "decisionboundary")
    ggplot2::ggsave(paste0(outprefix, "_IHWdecisionBoundaryPlot.png"), device = "png", path = "./")
  }

  ## Learn distance dependency (chicdiff.R:2004-2031)
  test <- ihwRes@df
  data.table::setDT(test)
  distLookup <- test[, list(avgLogDist = mean(log(covariate)), minLogDist = min(log(covariate)), maxLogDist = max(log(covariate))),
                     by = "group"]
  distLookup <- distLookup[!is.na(group), ]
  data.table::setkey(distLookup, group)
  if (distLookup[, !identical(as.integer(group), seq_along(group))]) stop("Assumption violated")
  w <- ihwRes@weights
  distLookup[, group := as.integer(group)]
  distLookup$avWeights <- rowSums(w) / ncol(w)
  distLookup$minLogDist[1] <- 0
  distLookup$maxLogDist[nrow(distLookup)] <- Inf
  message("Learned distance dependency")

  ## Apply to test data, on the device (chicdiff.R:2036-2049)
  out <- .hipApplyIHWweights(out, distLookup, device = gpu)
  message("applied to test data")

  if (diffbaitPlot == TRUE) {
    sel <- order(out$weighted_padj)
    baits <- sample(head(unique(out[sel]$baitID), 100), 4)
    plotDiffBaits(output = out, countput = countput, baitmapfile = baitmapfile, baits = baits)
    cowplot::ggsave2(paste0(outprefix, "_diffbaitPlot", ".", device), device = device, path = "./")
    dev.off()
  }
  saveRDS(out, paste0(outprefix, "_results", suffix, ".Rds"))
  out
}
