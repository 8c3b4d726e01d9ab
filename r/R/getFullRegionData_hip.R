## getFullRegionData_hip.R -- the step before DESeq2Wrap() with the MI355X backend behind it (SURVEY.md §8 f2, a1, a3).
##
## getFullRegionDataHip(chicdiff.settings, RU, is_control = FALSE, suffix = "") does what the reference's
## getFullRegionData1() (chicdiff.R:577-945) does for the chinput case, but never builds the long "recast" table
## (one row per region, fragment and sample: 176 M rows at 2 M regions x 8 samples).  It returns a
## "chicdiffHipRegionData" object that DESeq2Wrap() (r/R/DESeq2Wrap_hip.R) takes in place of FullRegionData:
## per-sample fragment columns N and FullMean, resident on the device in (regionID, otherEndID) order, plus the
## offsets of each region's first fragment.
##
##   reference step                                              here
##   fread(chinput); setkey; x[J(baits)]        chicdiff.R:828-831   chicdiff_hip_chinput_table (host threads + device sort)
##   merge(x, temp, all.x = TRUE); N[NA] <- 0   chicdiff.R:843-858   chicdiff_hip_count_join
##   s_j / s_i / tblb / tlb / Tmean collection  chicdiff.R:656-692   .hipBackgroundTables (R: reads the Chicago objects)
##   .chicEstimateDistFun, .estimateBMean       chicdiff.R:538-573, 695-702   R lm() refit, then chicdiff_hip_fragment_background
##   FullMean := Bmean + Tmean                  chicdiff.R:896       chicdiff_hip_fragment_background
##
## NOT run in this repository (no R in the authoring image or on the GPU box, SURVEY.md §0); the device routines it
## calls are tested through the same C-ABI from Python (tests/test_gpu_parity.py: count table, count join, fragment
## background on the reference's chr19 geometry, chinput ingestion).

.hipCall <- function(name, ...) .Call(name, ..., PACKAGE = "chicdiffhip")

## dense lookup tables over fragment IDs 1..nid for one Chicago data set `x` (a data.table with the columns of a
## chicagoData@x): first s_j / tblb per bait, first s_i / tlb per other end (chicdiff.R:656-672), the Tmean of every
## (tblb, tlb) pair (:676-681) and the distance function's ten numbers (:538-573)
.hipBackgroundTables <- function(x, nid, levB, levL) {
  sj <- rep(NA_real_, nid); si <- rep(NA_real_, nid)
  tblb <- rep(NA_integer_, nid); tlb <- rep(NA_integer_, nid)
  b <- x[, list(s_j = s_j[1L], tblb = tblb[1L]), by = "baitID"]
  b <- b[baitID >= 1L & baitID <= nid]
  sj[b$baitID] <- b$s_j
  tblb[b$baitID] <- match(as.character(b$tblb), levB)
  o <- x[, list(s_i = s_i[1L], tlb = tlb[1L]), by = "otherEndID"]
  o <- o[otherEndID >= 1L & otherEndID <= nid]
  si[o$otherEndID] <- o$s_i
  tlb[o$otherEndID] <- match(as.character(o$tlb), levL)
  tm <- x[!is.na(tblb) & !is.na(tlb), list(Tmean = Tmean[1L]), by = c("tblb", "tlb")]
  Tm <- matrix(NA_real_, nrow = length(levL), ncol = length(levB))   # tlb fastest: the library's [tblb][tlb] layout
  Tm[cbind(match(as.character(tm$tlb), levL), match(as.character(tm$tblb), levB))] <- tm$Tmean
  p <- .chicEstimateDistFun(x)                                        # the reference's own function (lm() on ~ 75 bins)
  list(sj = sj, si = si, tblb = tblb, tlb = tlb, Tmean = Tm,
       distfun = as.double(c(p$cubicFit[1:4], p$head.coef, p$tail.coef, p$obs.min, p$obs.max)))
}

getFullRegionDataHip <- function(chicdiff.settings, RU, is_control = FALSE, suffix = "") {
  countData <- chicdiff.settings[["countData"]]
  chicagoData <- chicdiff.settings[["chicagoData"]]
  rmapfile <- chicdiff.settings[["rmapfile"]]
  device <- if (is.null(chicdiff.settings[["device"]])) 0L else as.integer(chicdiff.settings[["device"]])
  if (is.null(countData)) stop("getFullRegionDataHip: the chinput files (countData) are required")
  targetRDSorRDAFiles <- unlist(chicagoData)
  targetChFiles <- unlist(countData)
  S <- length(targetChFiles)
  if (length(targetRDSorRDAFiles) != S) stop("getFullRegionDataHip: one Chicago data set per chinput file expected")
  condition <- rep(names(chicagoData), sapply(chicagoData, length))   # chicdiff.R:921-923

  ## RU in (regionID, otherEndID) order: a region's fragments are consecutive and ascending, as the window sums need
  ru <- RU[order(regionID, otherEndID)]
  ids <- unique(ru$regionID)
  if (!identical(as.integer(ids), seq_along(ids))) stop("RU: regionID must be 1..n without gaps")
  nfrag <- nrow(ru); n <- length(ids)
  region_ptr <- as.double(c(match(ids, ru$regionID) - 1L, nfrag))
  baits <- sort(unique(ru$baitID))                                     # chicdiff.R:775

  ctx <- .hipContext(device)
  dBait <- .hipCall("chicdiff_hip_upload", ctx, as.integer(ru$baitID))
  dOE <- .hipCall("chicdiff_hip_upload", ctx, as.integer(ru$otherEndID))
  on.exit({ .hipCall("chicdiff_hip_release", dBait); .hipCall("chicdiff_hip_release", dOE) }, add = TRUE)

  ## 2) read counts: one key table per replicate, joined onto RU straight into column i of the fragment matrix
  fragN <- .hipCall("chicdiff_hip_alloc", ctx, "integer", as.double(nfrag) * S)
  for (i in seq_len(S)) {
    message("Reading count data for ", names(targetChFiles)[i])
    tab <- .hipCall("chicdiff_hip_chinput_table", ctx, targetChFiles[i], as.integer(baits))
    .hipCall("chicdiff_hip_count_join", ctx, dBait, dOE, tab, fragN, as.double(i - 1L))
    .hipCall("chicdiff_hip_release", tab$keys); .hipCall("chicdiff_hip_release", tab$vals)
  }

  ## 1) interaction parameters: the per-fragment tables of every Chicago data set, then Bmean / Tmean / FullMean of
  ##    every RU row on the device
  rmap <- data.table::fread(rmapfile)
  data.table::setnames(rmap, c("chr", "start", "end", "ID"))
  nid <- max(rmap$ID)
  midsum <- rep(NA_real_, nid); midsum[rmap$ID] <- as.double(rmap$start) + as.double(rmap$end)
  xs <- vector("list", S)
  dispersions <- numeric(S)
  for (i in seq_len(S)) {
    message("\nReading Chicago dataset ", i, " of ", S, " : ", names(targetRDSorRDAFiles)[i])
    x <- readRDSorRDA(targetRDSorRDAFiles[i])
    if ("chicagoData" %in% class(x)) { dispersions[i] <- x@params$dispersion; x <- data.table::as.data.table(x@x) }
    else { dispersions[i] <- attributes(x)$dispersion; data.table::setDT(x) }
    xs[[i]] <- x[, c("baitID", "otherEndID", "s_j", "s_i", "tblb", "tlb", "Tmean", "distbin", "refBinMean"), with = FALSE]
  }
  levB <- sort(unique(unlist(lapply(xs, function(x) as.character(x$tblb[!is.na(x$tblb)])))))
  levL <- sort(unique(unlist(lapply(xs, function(x) as.character(x$tlb[!is.na(x$tlb)])))))
  tabs <- lapply(xs, .hipBackgroundTables, nid = nid, levB = levB, levL = levL)
  bg <- .hipCall("chicdiff_hip_fragment_background", ctx, dBait, dOE, 1L, midsum,
                 unlist(lapply(tabs, function(t) t$sj)), unlist(lapply(tabs, function(t) t$si)),
                 unlist(lapply(tabs, function(t) t$tblb)), unlist(lapply(tabs, function(t) t$tlb)),
                 array(unlist(lapply(tabs, function(t) t$Tmean)), dim = c(length(levL), length(levB), S)),
                 unlist(lapply(tabs, function(t) t$distfun)), S)
  .hipCall("chicdiff_hip_release", bg$Bmean); .hipCall("chicdiff_hip_release", bg$Tmean)

  structure(list(samples = names(targetRDSorRDAFiles), condition = condition, S = S, n = n, fragN = fragN,
                 fragFullMean = bg$FullMean, region_ptr = region_ptr, dispersions = dispersions, is_control = is_control),
            class = "chicdiffHipRegionData")
}
