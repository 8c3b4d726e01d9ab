## DESeq2Wrap_hip.R -- the MI355X path of Chicdiff's DESeq2Wrap().
##
## The exported function of the reference (chicdiff.R:1494),
##     DESeq2Wrap(chicdiff.settings, RU, FullRegionData, suffix = "", theta = NULL)
## stays the reference's own: r/patches/chicdiff_hip.patch adds, after its argument handling and warnings (:1496-1521), the call
##     if (identical(chicdiff.settings[["backend"]], "hip")) return(.DESeq2WrapHip(..., theta = theta, norm = norm))
## .DESeq2WrapHip() below replaces the DESeq2 hand-off of chicdiff.R:1540-1691 (+ results(), :1720-1750) by libchicdiff_hip.so
## through the .Call routines of r/src/chicdiff_hip_shim.c, with the reference's messages and its returned data.table (columns
## and order of chicdiff.R:1752-1757, attr "theta" as :1759-1760).  New optional settings: backend ("hip"), hipDevice (GPU
## index, default 0; NOT `device`, which is the reference's plot device "png", chicdiff.R:20), trendFallback ("mean").
##
## NOT run in this repository: there is no R in the authoring image or on the GPU box (SURVEY.md §0).  The tested
## twin with the same control flow is chicdiff_amd/deseq2wrap.py (pandas standing in for data.table).

.hipEnv <- new.env(parent = emptyenv())

## one context per R process and device (one R process per GPU when sharding, INTEGRATION.md §4)
.hipContext <- function(device = 0L) {
  key <- paste0("ctx", device)
  if (is.null(.hipEnv[[key]]))
    .hipEnv[[key]] <- .Call("chicdiff_hip_open", as.integer(device), PACKAGE = "chicdiffhip")
  .hipEnv[[key]]
}

## character condition -> factor: alphabetical levels, the first is the reference level (DESeqDataSetFromMatrix)
.hipGroup <- function(condition) {
  lev <- sort(unique(as.character(condition)))
  if (length(lev) != 2L) stop("design ~ condition needs exactly two conditions, got: ", paste(lev, collapse = ", "))
  as.integer(as.character(condition) == lev[2L])
}

## qf(.99, p, m - p), or NA when DESeq2 would skip the Cook's cutoff (no group with >= 3 replicates)
.hipCooksCutoff <- function(group) {
  m <- length(group); p <- 2L
  if (m > p && max(table(group)) >= 3L) stats::qf(.99, p, m - p) else NA_real_
}

.hipNA <- function(fit) {
  for (k in c("baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "padj", "dispGeneEst", "dispFit", "dispersion",
              "deviance", "maxCooks"))
    fit[[k]][is.nan(fit[[k]])] <- NA_real_   # all-zero rows / filtered rows: DESeq2 reports NA
  fit
}

## A failed parametric trend is replaced by the local regression inside the library, as estimateDispersionsFit does
## (status bit 16); DESeq2's message is repeated here.  Only when not even the local fit exists (fewer than four rows
## with a usable dispersion estimate, status bit 1) is there a choice to make: options(chicdiff.hip.trendFallback =
## "mean") refits with DESeq2's other documented alternative, fitType = "mean"; the default is to stop.
.hipTrendFailed <- function(fit) bitwAnd(fit$status, 1L) != 0L
.hipFitWithFallback <- function(run) {
  fit <- run(0L)
  if (bitwAnd(fit$status, 16L) != 0L)
    message("-- note: fitType='parametric', but the dispersion trend was not well captured by the\n",
            "   function: y = a/x + b, and a local regression fit was automatically substituted.\n",
            "   specify fitType='local' or 'mean' to avoid this message next time.")
  if (.hipTrendFailed(fit)) {
    if (!identical(getOption("chicdiff.hip.trendFallback"), "mean"))
      stop("no dispersion trend could be fitted (parametric and local fits both failed): ",
           "set options(chicdiff.hip.trendFallback = \"mean\") to refit with fitType = \"mean\"")
    message("-- note: no dispersion trend could be fitted; fitType = \"mean\" was used instead.")
    fit <- run(1L)
  }
  fit
}

## estimateDispersions + nbinomWaldTest + results() for given normalisation factors: what a caller outside
## DESeq2Wrap() uses (and tools/make_golden.R compares with DESeq2 itself)
DESeq2Hip <- function(regionDataMatrix, normFactors, condition, device = 0L, alpha = 0.1, dispPriorVar = NA_real_) {
  storage.mode(regionDataMatrix) <- "integer"   # n x S, column-major = sample-major
  storage.mode(normFactors) <- "double"
  group <- .hipGroup(condition)
  .hipNA(.hipFitWithFallback(function(fitType)
    .Call("chicdiff_hip_fit", .hipContext(device), regionDataMatrix, normFactors, group, as.double(dispPriorVar), fitType,
          .hipCooksCutoff(group), as.double(alpha), as.double(nrow(regionDataMatrix)), ncol(regionDataMatrix),
          PACKAGE = "chicdiffhip")))
}

## long "recast" table -> dense per-sample fragment columns in (regionID, otherEndID) order + region offsets.
## setkey(fragData, otherEndID) and by = c("baitID", "regionID", "sample") of chicdiff.R:1526, :1540-1547: inside a
## region the fragments are added in ascending otherEndID order.
.hipDenseFragments <- function(FullRegionData) {
  fd <- data.table::copy(FullRegionData)        # as the reference: the caller's table must not change
  data.table::setkeyv(fd, "otherEndID")
  samples <- unique(fd$sample)
  S <- length(samples)
  condition <- fd$condition[seq_len(S)]   # colData, chicdiff.R:1556
  if (!identical(as.character(fd$sample[seq_len(S)]), as.character(samples)))
    stop("FullRegionData: the first rows do not hold one row per sample (recast layout expected)")
  if (anyNA(fd$N)) stop("FullRegionData: NA counts")
  ord <- order(match(fd$sample, samples), fd$regionID, fd$otherEndID)
  nfrag <- nrow(fd) %/% S
  if (nfrag * S != nrow(fd)) stop("FullRegionData: samples do not cover the same (regionID, otherEndID) rows")
  region <- matrix(fd$regionID[ord], ncol = S)
  oe <- matrix(fd$otherEndID[ord], ncol = S)
  if (any(region != region[, 1L]) || any(oe != oe[, 1L]))
    stop("FullRegionData: samples do not cover the same (regionID, otherEndID) rows")
  ids <- unique(region[, 1L])
  if (!identical(as.integer(ids), seq_along(ids)))
    stop("identical(1:nrow(annoData), annoData$regionID) is not TRUE")  # the reference's stopifnot, chicdiff.R:1717
  list(samples = samples, condition = condition, S = S, n = length(ids),
       fragN = matrix(as.integer(fd$N[ord]), ncol = S),
       fragFullMean = matrix(as.double(fd$FullMean[ord]), ncol = S),
       region_ptr = as.double(c(match(ids, region[, 1L]) - 1L, nfrag)))
}

## annotation columns of the output table, chicdiff.R:1699-1717
.hipAnnotation <- function(RU, rmapfile, n) {
  rmap <- .hipReadRmap(rmapfile)
  ru <- RU[, list(baitID = baitID[1L], minOE = min(otherEndID), maxOE = max(otherEndID)), by = "regionID"]
  data.table::setkey(ru, regionID)
  lo <- match(ru$minOE, rmap$ID); hi <- match(ru$maxOE, rmap$ID); b <- match(ru$baitID, rmap$ID)
  keep <- !is.na(lo) & !is.na(hi) & !is.na(b)   # merge() drops regions whose fragments are not on the map
  anno <- data.table::data.table(baitID = ru$baitID, maxOE = ru$maxOE, minOE = ru$minOE, regionID = ru$regionID,
                                 OEchr = rmap$chr[lo], OEstart = rmap$start[lo], OEend = rmap$end[hi],
                                 baitchr = rmap$chr[b], baitstart = rmap$start[b], baitend = rmap$end[b])[keep]
  stopifnot(identical(seq_len(nrow(anno)), as.integer(anno$regionID)), nrow(anno) == n)
  anno
}

## Entered from the reference's DESeq2Wrap() AFTER its own argument handling (chicdiff.R:1496-1521: theta from the settings,
## the unknown-norm stop, the two theta = 0 / 1 warnings that rewrite `norm`): r/patches/chicdiff_hip.patch adds the call
## there, so `theta` and `norm` arrive resolved and none of those statements is restated here.
.DESeq2WrapHip <- function(chicdiff.settings, RU, FullRegionData, suffix = "", theta = NULL, norm = chicdiff.settings[["norm"]]) {

  Grid <- chicdiff.settings[["theta_grid"]]
  rmapfile <- chicdiff.settings[["rmapfile"]]
  saveAux <- chicdiff.settings[["saveAuxData"]]
  outprefix <- chicdiff.settings[["outprefix"]]
  device <- .hipDeviceIndex(chicdiff.settings)   # the new key `hipDevice` (r/R/getFullRegionData_hip.R); never `device`

  ctx <- .hipContext(device)
  ## either the reference's long "recast" table, or the device-resident fragment block of getFullRegionDataHip()
  fr <- if (inherits(FullRegionData, "chicdiffHipRegionData")) FullRegionData else .hipDenseFragments(FullRegionData)
  n <- fr$n; S <- fr$S
  group <- .hipGroup(fr$condition)

  ## window sums (chicdiff.R:1540-1556): the count and FullMean matrices stay on the device from here on
  ws <- .Call("chicdiff_hip_window_sums", ctx, fr$fragN, if (norm != "standard") fr$fragFullMean else NULL,
              fr$region_ptr, S, PACKAGE = "chicdiffhip")
  on.exit({ .Call("chicdiff_hip_release", ws$N, PACKAGE = "chicdiffhip")
            if (!is.null(ws$FullMean)) .Call("chicdiff_hip_release", ws$FullMean, PACKAGE = "chicdiffhip") }, add = TRUE)

  cooks <- .hipCooksCutoff(group)
  fitWith <- function(fullMean, tt)   # size factors -> sc -> estimateDispersions -> nbinomWaldTest -> results()
    .hipNA(.hipFitWithFallback(function(fitType)
      .Call("chicdiff_hip_wald_test", ctx, ws$N, fullMean, group, as.double(tt), NA_real_, fitType, cooks, 0.1,
            as.double(n), S, PACKAGE = "chicdiffhip")))

  if (identical(norm, "standard")) {  # model 1: size factors only (chicdiff.R:1572-1575)
    fit <- fitWith(NULL, NA_real_)
    label <- "Standard DESeq2 normalisation"
  }
  if (identical(norm, "fullmean")) {  # model 3: normFactorsM3 (chicdiff.R:1598-1604)
    fit <- fitWith(ws$FullMean, NA_real_)
    label <- "Chicago full mean-based normalisation"
  }
  if (identical(norm, "combined")) {  # model 5: sc(theta) (chicdiff.R:1612-1674)
    tt <- theta
    if (!length(tt)) {               # NULL: scan the grid
      message("Optimising scaling factors...")
      nullSizeFactors <- .Call("chicdiff_hip_size_factors", ctx, ws$N, as.double(n), S, PACKAGE = "chicdiffhip")
      deviances <- .Call("chicdiff_hip_theta_grid", ctx, ws$N, ws$FullMean, nullSizeFactors, as.double(Grid), as.double(n), S,
                         PACKAGE = "chicdiffhip")
      deviances[is.nan(deviances)] <- NA_real_   # sum() without na.rm over an all-zero row, chicdiff.R:1647
      message("Total deviances by theta (Fullmean --> Standard):")
      cat(sprintf("%f", deviances), "\n", file = stderr())
      tt <- Grid[which(deviances == min(deviances)[1])]
    }
    message("Theta=", tt)
    ## A deliberate stop (INTEGRATION.md 2): the reference carries on with whatever which() returned — numeric(0) when an
    ## all-zero region made every deviance NA (sum() without na.rm, chicdiff.R:1647), several values on a tie — and fails later, in
    ## `normFactorsM3*(1-tt)` (:1666), with R's own "non-conformable" / recycling message.  A .Call cannot be handed a
    ## zero-length or longer theta, so the same situation is reported here, in words.
    if (length(tt) != 1L) stop("theta grid: no unique minimum of the total deviance (an all-zero region makes every deviance NA)")
    fit <- fitWith(ws$FullMean, tt)
    label <- "combined normalisation"
  }

  message("Processing model output")
  annoData <- .hipAnnotation(RU, rmapfile, n)

  message(label, ": # unweighted interactions with padj<0.05: ", sum(fit$padj < 0.05 & !is.na(fit$padj)))
  if (saveAux == TRUE) {
    ## the reference saves the DESeqDataSet here; the GPU path has no S4 object, so the fit's own list is saved
    message(if (norm == "combined") "Saving the final DESeq object" else "Saving the DESeq object")
    saveRDS(fit, paste0(outprefix, "_DESeqObj", suffix, ".Rds"))
  }

  out <- cbind(data.table::data.table(baseMean = fit$baseMean, log2FoldChange = fit$log2FoldChange, lfcSE = fit$lfcSE,
                                      stat = fit$stat, pvalue = fit$pvalue, padj = fit$padj), annoData)
  if (norm == "combined") attributes(out)$theta <- tt
  out
}
