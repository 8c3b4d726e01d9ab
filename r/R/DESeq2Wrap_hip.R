## DESeq2Wrap_hip.R -- drop-in body for the DESeq2 hand-off inside Chicdiff's DESeq2Wrap().
##
## NOT run in this repository (no R in the authoring image or on the GPU box).  It shows the
## reference-side binding: the caller-visible function keeps the reference signature
##     DESeq2Wrap(chicdiff.settings, RU, FullRegionData, suffix = "", theta = NULL)   (chicdiff.R:1494)
## and only the block chicdiff.R:1557-1691 (+ results(), :1720-1750) is swapped for .Call()s into
## r/src/chicdiff_hip_shim.c when chicdiff.settings[["backend"]] == "hip"; otherwise the reference
## DESeq2 path runs unchanged.  The Python mirror chicdiff_amd/deseq2wrap.py is the tested twin.

.hipFit <- function(counts, nf, condition, dispPriorVar = NA_real_) {
  storage.mode(counts) <- "integer"            # n x S, column-major = sample-major
  storage.mode(nf) <- "double"
  lev <- sort(unique(as.character(condition)))  # character -> factor: alphabetical, first = reference
  stopifnot(length(lev) == 2L)
  group <- as.integer(as.character(condition) == lev[2L])
  r <- .Call("chicdiff_hip_fit", counts, nf, group, as.double(dispPriorVar), PACKAGE = "chicdiffhip")
  for (k in c("baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "deviance"))
    r[[k]][is.nan(r[[k]])] <- NA_real_          # all-zero rows: DESeq2 reports NA
  r
}

## residual d.f. <= 3 (e.g. 2 vs 2): DESeq2's prior variance is a Monte-Carlo match drawn from R's session
## RNG.  The library runs the same matching from a fixed-seed stream of its own (status bit 2), which is the
## default; set options(chicdiff.hip.priorvar = "DESeq2") to have it computed HERE with DESeq2's own code
## (and R's RNG) from the GPU's gene-wise estimates and handed back through chicdiff_nbglm_opts.dispPriorVar,
## exactly the argument DESeq2 exposes.
.hipPriorVar <- function(fit, modelMatrix, minDisp = 1e-8) {
  m <- nrow(modelMatrix); p <- ncol(modelMatrix)
  if (!((m - p) <= 3 && m > p)) return(NA_real_)
  dds <- DESeq2::makeExampleDESeqDataSet(n = length(fit$dispGeneEst), m = m)
  S4Vectors::mcols(dds)$dispGeneEst <- fit$dispGeneEst
  S4Vectors::mcols(dds)$dispFit <- fit$dispFit
  S4Vectors::mcols(dds)$allZero <- is.na(fit$dispGeneEst)
  DESeq2::estimateDispersionsPriorVar(dds, modelMatrix = modelMatrix)
}

DESeq2Hip <- function(regionDataMatrix, normFactors, condition) {
  fit <- .hipFit(regionDataMatrix, normFactors, condition)
  X <- stats::model.matrix(~ condition, data.frame(condition = factor(condition)))
  pv <- if (identical(getOption("chicdiff.hip.priorvar"), "DESeq2")) .hipPriorVar(fit, X) else NA_real_
  if (!is.na(pv)) fit <- .hipFit(regionDataMatrix, normFactors, condition, dispPriorVar = pv)
  if (bitwAnd(fit$status, 1L)) stop("parametric dispersion trend failed (DESeq2 would use a local fit)")
  fit
}
