## getFullRegionData_hip.R -- the step before DESeq2Wrap() with the MI355X backend behind it (SURVEY.md §8 f2, a1, a3).
##
## The device path of the exported function of the reference (chicdiff.R:1460-1478),
##     getFullRegionData(chicdiff.settings, RU, RUcontrol, suffix = "")  ->  .getFullRegionDataHip(same arguments)
## With chicdiff.settings[["backend"]] == "hip" it returns list(test, control, countput) like the reference, but the
## first two entries are "chicdiffHipRegionData" blocks instead of the long "recast" tables (one row per region,
## fragment and sample: 176 M rows at 2 M regions x 8 samples, 3.5 G at 20 M x 16): per-sample fragment columns N and
## FullMean resident on the device in (regionID, otherEndID) order, the offsets of each region's first fragment, and
## the per-region avDist that IHWcorrection() otherwise takes from the long table (chicdiff.R:1965).  DESeq2Wrap()
## (r/R/DESeq2Wrap_hip.R) and IHWcorrection() (r/R/post_hip.R) accept the blocks in place of the tables.  Every
## Chicago data set and every chinput file is read ONCE for both universes -- what parallel = TRUE
## (getFullRegionData2, chicdiff.R:948-1456) does in the reference; the result does not depend on it.  With any other
## backend the reference's own function body runs: the patch (r/patches/chicdiff_hip.patch) adds one line at its top.
##
##   reference step                                              here
##   fread(chinput); setkey; x[J(baits)]        chicdiff.R:828-831   chicdiff_hip_chinput_table (host threads + device sort)
##   merge(x, temp, all.x = TRUE); N[NA] <- 0   chicdiff.R:843-858   chicdiff_hip_count_join_multi (all replicates; _count_join: one)
##   no chinput: Reduce(merge, tempForCounts)   chicdiff.R:742-747, 774-807   chicdiff_hip_count_table + chicdiff_hip_count_join_inner
##   s_j / s_i / tblb / tlb / Tmean collection  chicdiff.R:656-692   .hipBackgroundTables (R: reads the Chicago objects)
##   .chicEstimateDistFun, .estimateBMean       chicdiff.R:538-573, 695-702   R lm() refit, then chicdiff_hip_fragment_background
##   FullMean := Bmean + Tmean                  chicdiff.R:896       chicdiff_hip_fragment_background
##   distSign of CountOut, mean by regionID     chicdiff.R:868-882, 1965   chicdiff_hip_region_avdist
##   countput                                   chicdiff.R:708-735, 754-769   the reference's own data.table statements
##
## NOT run in this repository (no R in the authoring image or on the GPU box, SURVEY.md §0); the tested twin with the
## same control flow is chicdiff_amd/pipeline.py:getFullRegionData (tests/test_gpu_parity.py::test_chicdiffPipeline_*).

.hipCall <- function(name, ...) .Call(name, ..., PACKAGE = "chicdiffhip")

## the restriction map as Chicago's .readRmap gives it: four columns chr, start, end, ID
.hipReadRmap <- function(path) {
  m <- data.table::fread(path)
  data.table::setnames(m, c("chr", "start", "end", "ID"))
  m
}

## GPU index of this R process: the NEW optional key `hipDevice` (default 0).  Never `device`: that key is the
## reference's plot device ("png", chicdiff.R:20, used at :1960 and :2058).
.hipDeviceIndex <- function(chicdiff.settings) {
  d <- chicdiff.settings[["hipDevice"]]
  if (is.null(d)) return(0L)
  d <- suppressWarnings(as.integer(d))
  if (length(d) != 1L || is.na(d) || d < 0L) stop("chicdiff.settings[[\"hipDevice\"]] must be one non-negative GPU index")
  d
}

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

## countput (chicdiff.R:708-735 per replicate, :754-769 afterwards): what plotDiffBaits() draws.  `xs`: the Chicago
## tables, `conditions`: one label per replicate, `rmap`: data.table chr, start, end, ID.
.hipCountput <- function(xs, conditions, rmap, outprefix) {
  mid <- data.table::data.table(otherEndID = rmap$ID, midpoint = (rmap$start + rmap$end) / 2)
  countput <- lapply(unique(conditions), function(cond) {
    parts <- lapply(which(conditions == cond), function(i) {
      x <- xs[[i]][!is.na(distSign)]
      sc <- if ("newScore" %in% names(x)) "newScore" else "score"
      x <- x[, c("baitID", "otherEndID", "N", "Bmean", sc), with = FALSE]
      data.table::setnames(x, sc, "score")
      merge(x, mid, by = "otherEndID")
    })
    z <- data.table::rbindlist(parts)
    z <- z[, list(Nav = mean(N), Bav = mean(Bmean), score = max(score), midpoint = midpoint[1L]), by = c("baitID", "otherEndID")]
    z[, condition := cond]
    z
  })
  countput <- data.table::rbindlist(countput)
  data.table::setnames(countput, "midpoint", "oeID_mid")
  saveRDS(countput, paste0(outprefix, "_countput.Rds"))
  countput
}

## RU in (regionID, otherEndID) order + the offsets of each region's first row: what the device kernels consume
.hipRegionCSR <- function(RU) {
  ru <- RU[order(regionID, otherEndID)]
  ids <- unique(ru$regionID)
  if (!identical(as.integer(ids), seq_along(ids)))
    stop("identical(1:nrow(annoData), annoData$regionID) is not TRUE")  # what the reference's stopifnot says later (chicdiff.R:1717)
  list(baitID = as.integer(ru$baitID), otherEndID = as.integer(ru$otherEndID), n = length(ids), nfrag = nrow(ru),
       region_ptr = as.double(c(match(ids, ru$regionID) - 1L, nrow(ru))))
}

.getFullRegionDataHip <- function(chicdiff.settings, RU, RUcontrol, suffix = "") {
  countData <- chicdiff.settings[["countData"]]
  chicagoData <- chicdiff.settings[["chicagoData"]]
  rmapfile <- chicdiff.settings[["rmapfile"]]
  outprefix <- chicdiff.settings[["outprefix"]]
  targetRDSorRDAFiles <- unlist(chicagoData)
  S <- length(targetRDSorRDAFiles)
  haveChinput <- !is.null(countData) && !all(is.na(unlist(countData)))
  if (haveChinput) {
    targetChFiles <- unlist(countData)
    if (length(targetChFiles) != S) stop("Must provide the same number of RDS/RDA files as chinputs")
  }
  conditions <- rep(names(chicagoData), sapply(chicagoData, length))   # chicdiff.R:921-923
  ctx <- .hipContext(.hipDeviceIndex(chicdiff.settings))

  rmap <- .hipReadRmap(rmapfile)
  nid <- max(rmap$ID)
  midsum <- rep(NA_real_, nid); midsum[rmap$ID] <- as.double(rmap$start) + as.double(rmap$end)
  chrcode <- rep(-1L, nid); chrcode[rmap$ID] <- as.integer(factor(rmap$chr)) - 1L

  ## 1) the Chicago data sets, each read once: dispersion, per-fragment tables, countput, (no chinput:) the counts
  xs <- vector("list", S)
  dispersions <- numeric(S)
  for (i in seq_len(S)) {
    message("\nReading Chicago dataset ", i, " of ", S, " : ", names(targetRDSorRDAFiles)[i])
    x <- readRDSorRDA(targetRDSorRDAFiles[i])
    if ("chicagoData" %in% class(x)) { dispersions[i] <- x@params$dispersion; x <- data.table::as.data.table(x@x) }
    else { dispersions[i] <- attributes(x)$dispersion; data.table::setDT(x) }
    data.table::setkey(x, baitID, otherEndID)                          # chicdiff.R:630: "first per bait / other end" is in this order
    xs[[i]] <- x
  }
  levB <- sort(unique(unlist(lapply(xs, function(x) as.character(x$tblb[!is.na(x$tblb)])))))
  levL <- sort(unique(unlist(lapply(xs, function(x) as.character(x$tlb[!is.na(x$tlb)])))))
  tabs <- lapply(xs, .hipBackgroundTables, nid = nid, levB = levB, levL = levL)
  message("Saving counts\n")
  countput <- .hipCountput(xs, conditions, rmap, outprefix)

  ## 2) one key table per replicate, restricted to the baits of both universes (chicdiff.R:775, 828-831)
  baits <- sort(unique(c(RU$baitID, RUcontrol$baitID)))
  flags <- .hipCall("chicdiff_hip_bait_flags", ctx, as.integer(baits))   # built and uploaded once, not once per replicate
  tables <- vector("list", S)
  if (haveChinput) {
    for (i in seq_len(S)) {
      message("Reading count data for ", names(targetChFiles)[i])
      tables[[i]] <- .hipCall("chicdiff_hip_chinput_table", ctx, targetChFiles[i], flags)
    }
  } else {
    message("Reconstructing countData")                                # chicdiff.R:742-747, 774-787
    for (i in seq_len(S))
      tables[[i]] <- .hipCall("chicdiff_hip_count_table", ctx, as.integer(xs[[i]]$baitID), as.integer(xs[[i]]$otherEndID),
                              as.integer(xs[[i]]$N), flags)
  }
  .hipCall("chicdiff_hip_release", flags)

  ## 3) per universe: counts, Bmean + Tmean = FullMean, avDist -- all of them stay on the device
  block <- function(ru, is_control) {
    message(if (!is_control) "Reading data for significant interactions" else "\nReading data for control interactions")
    csr <- .hipRegionCSR(ru)
    dBait <- .hipCall("chicdiff_hip_upload", ctx, csr$baitID)
    dOE <- .hipCall("chicdiff_hip_upload", ctx, csr$otherEndID)
    on.exit({ .hipCall("chicdiff_hip_release", dBait); .hipCall("chicdiff_hip_release", dOE) }, add = TRUE)
    if (haveChinput) {
      ## the loop over the replicates of chicdiff.R:843-858 as ONE pass over the RU rows (every replicate's merge() read them again)
      fragN <- .hipCall("chicdiff_hip_count_join_multi", ctx, dBait, dOE, tables)
    } else {
      message("Merging countData")                                     # chicdiff.R:789-805: inner merge over the replicates
      fragN <- .hipCall("chicdiff_hip_count_join_inner", ctx, dBait, dOE, tables)
    }
    bg <- .hipCall("chicdiff_hip_fragment_background", ctx, dBait, dOE, 1L, midsum,
                   unlist(lapply(tabs, function(t) t$sj)), unlist(lapply(tabs, function(t) t$si)),
                   unlist(lapply(tabs, function(t) t$tblb)), unlist(lapply(tabs, function(t) t$tlb)),
                   array(unlist(lapply(tabs, function(t) t$Tmean)), dim = c(length(levL), length(levB), S)),
                   unlist(lapply(tabs, function(t) t$distfun)), S)
    .hipCall("chicdiff_hip_release", bg$Bmean); .hipCall("chicdiff_hip_release", bg$Tmean)
    avDist <- .hipCall("chicdiff_hip_region_avdist", ctx, dBait, dOE, csr$region_ptr, 1L, midsum, chrcode)
    structure(list(samples = names(targetRDSorRDAFiles), condition = conditions, S = S, n = csr$n, fragN = fragN,
                   fragFullMean = bg$FullMean, region_ptr = csr$region_ptr, avDist = avDist, dispersions = dispersions,
                   is_control = is_control),
              class = "chicdiffHipRegionData")
  }
  out <- list(block(RU, FALSE), block(RUcontrol, TRUE), countput)
  for (t in tables) { .hipCall("chicdiff_hip_release", t$keys); .hipCall("chicdiff_hip_release", t$vals) }
  out
}

## (entered from the first line of the reference's getFullRegionData() when backend == "hip": r/patches/chicdiff_hip.patch)
