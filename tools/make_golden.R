#!/usr/bin/env Rscript
## make_golden.R -- pins the oracle (and through it the HIP library) to DESeq2 itself, on any box with R + DESeq2.
##
## Own code (no reference file is read): for every <tag> written by tools/export_synth.py it runs exactly the call
## sequence Chicdiff's DESeq2Wrap() makes (chicdiff.R:1557-1562, 1671-1674, 1739) --
##   DESeqDataSetFromMatrix -> normalizationFactors<- -> estimateDispersions -> nbinomWaldTest -> results
## -- and writes <out>/<tag>.deseq2.f64: a column-major double matrix with the columns named in <tag>.deseq2.cols
## plus <tag>.deseq2.scalars.txt (trend coefficients, varLogDispEsts, dispPriorVar, DESeq2 / R versions).
## tests/test_golden_deseq2.py compares the oracle (CPU) and the HIP library (GPU) with these files when they exist
## under tests/golden/deseq2/.  Also writes rng_check.txt: known draws after set.seed(2), the stream
## estimateDispersionsPriorVar uses (tests/test_r_rng.py holds the same numbers from the restated generators).
##
##   python tools/export_synth.py --out golden_inputs && Rscript tools/make_golden.R golden_inputs tests/golden/deseq2
## With a third argument "hip" it also runs the HIP backend through r/R/DESeq2Wrap_hip.R (DESeq2Hip) and prints the
## largest relative differences of log2FoldChange / pvalue / padj.

suppressPackageStartupMessages({ library(DESeq2) })
args <- commandArgs(trailingOnly = TRUE)
indir <- if (length(args) >= 1) args[1] else "golden_inputs"
outdir <- if (length(args) >= 2) args[2] else "tests/golden/deseq2"
withHip <- length(args) >= 3 && args[3] == "hip"
dir.create(outdir, recursive = TRUE, showWarnings = FALSE)

tags <- sub("\\.meta\\.txt$", "", list.files(indir, pattern = "\\.meta\\.txt$"))
for (tag in tags) {
  meta <- readLines(file.path(indir, paste0(tag, ".meta.txt")))
  dims <- as.integer(strsplit(meta[1], " ")[[1]]); n <- dims[1]; S <- dims[2]
  group <- as.integer(strsplit(meta[2], " ")[[1]])
  counts <- matrix(readBin(file.path(indir, paste0(tag, ".counts.i32")), "integer", n * S, size = 4), nrow = n)
  nf <- matrix(readBin(file.path(indir, paste0(tag, ".nf.f64")), "double", n * S, size = 8), nrow = n)
  rownames(counts) <- seq_len(n); colnames(counts) <- paste0("s", seq_len(S))
  intercept <- all(group == 0L)
  colData <- data.frame(condition = ifelse(group == 1L, "B", "A"))   # character -> factor, as chicdiff.R:1556-1559
  dds <- DESeqDataSetFromMatrix(countData = counts, colData = colData, design = if (intercept) ~ 1 else ~ condition)
  sf <- sizeFactors(estimateSizeFactors(dds))
  normalizationFactors(dds) <- nf
  dds <- estimateDispersions(dds)
  dds <- nbinomWaldTest(dds)
  mc <- mcols(dds)
  cols <- list(baseMean = mc$baseMean, baseVar = mc$baseVar, allZero = as.numeric(mc$allZero),
               dispGeneEst = mc$dispGeneEst, dispFit = mc$dispFit, dispMAP = mc$dispMAP, dispersion = mc$dispersion,
               dispOutlier = as.numeric(mc$dispOutlier), deviance = mc$deviance, betaConv = as.numeric(mc$betaConv),
               maxCooks = if (is.null(mc$maxCooks)) rep(NA_real_, n) else mc$maxCooks,
               intercept = mc$Intercept, interceptSE = mc$SE_Intercept)
  if (!intercept) {
    res <- results(dds)
    cols <- c(cols, list(log2FoldChange = res$log2FoldChange, lfcSE = res$lfcSE, stat = res$stat, pvalue = res$pvalue,
                         padj = res$padj, waldPvalue = mc$WaldPvalue_condition_B_vs_A))
  }
  writeBin(as.double(do.call(cbind, cols)), file.path(outdir, paste0(tag, ".deseq2.f64")), size = 8)
  writeLines(names(cols), file.path(outdir, paste0(tag, ".deseq2.cols")))
  df <- dispersionFunction(dds)
  writeLines(c(sprintf("n %d", n), sprintf("S %d", S), sprintf("group %s", paste(group, collapse = " ")),
               sprintf("sizeFactors %s", paste(sprintf("%.17g", sf), collapse = " ")),
               sprintf("trendCoef %.17g %.17g", attr(df, "coefficients")[1], attr(df, "coefficients")[2]),
               sprintf("varLogDispEsts %.17g", attr(df, "varLogDispEsts")),
               sprintf("dispPriorVar %.17g", attr(df, "dispPriorVar")),
               sprintf("fitType %s", attr(df, "fitType")),
               sprintf("sumDeviance %.17g", sum(mc$deviance)),
               sprintf("DESeq2 %s", as.character(packageVersion("DESeq2"))), sprintf("R %s", R.version.string)),
             file.path(outdir, paste0(tag, ".deseq2.scalars.txt")))
  file.copy(file.path(indir, paste0(tag, c(".counts.i32", ".nf.f64", ".meta.txt"))), outdir, overwrite = TRUE)  # inputs travel with the goldens
  message(tag, ": dispPriorVar ", attr(df, "dispPriorVar"), "  fitType ", attr(df, "fitType"))

  ## the same data with DESeq2's local-regression trend (localDispersionFit = locfit; also what DESeq2 substitutes when
  ## the parametric fit fails): pins oracle/locfit_oracle.c and the library's fitType = 2.  Written as tag "<tag>_local".
  ddl <- DESeqDataSetFromMatrix(countData = counts, colData = colData, design = if (intercept) ~ 1 else ~ condition)
  normalizationFactors(ddl) <- nf
  ddl <- nbinomWaldTest(estimateDispersions(ddl, fitType = "local"))
  ml <- mcols(ddl); dl <- dispersionFunction(ddl)
  lcols <- list(baseMean = ml$baseMean, allZero = as.numeric(ml$allZero), dispGeneEst = ml$dispGeneEst, dispFit = ml$dispFit,
                dispMAP = ml$dispMAP, dispersion = ml$dispersion)
  if (!intercept) lcols <- c(lcols, list(log2FoldChange = ml$condition_B_vs_A, waldPvalue = ml$WaldPvalue_condition_B_vs_A))
  ltag <- paste0(tag, "_local")
  writeBin(as.double(do.call(cbind, lcols)), file.path(outdir, paste0(ltag, ".deseq2.f64")), size = 8)
  writeLines(names(lcols), file.path(outdir, paste0(ltag, ".deseq2.cols")))
  writeLines(c(sprintf("n %d", n), sprintf("S %d", S), sprintf("group %s", paste(group, collapse = " ")),
               "trendCoef NaN NaN", sprintf("varLogDispEsts %.17g", attr(dl, "varLogDispEsts")),
               sprintf("dispPriorVar %.17g", attr(dl, "dispPriorVar")), sprintf("fitType %s", attr(dl, "fitType")),
               sprintf("locfit %s", as.character(packageVersion("locfit")))),
             file.path(outdir, paste0(ltag, ".deseq2.scalars.txt")))
  for (ext in c(".counts.i32", ".nf.f64", ".meta.txt"))
    file.copy(file.path(indir, paste0(tag, ext)), file.path(outdir, paste0(ltag, ext)), overwrite = TRUE)

  if (withHip && !intercept) {
    source("r/R/DESeq2Wrap_hip.R")
    dyn.load("r/src/chicdiff_hip_shim.so")
    fit <- DESeq2Hip(counts, nf, colData$condition)
    rel <- function(a, b) { ok <- !is.na(a) & !is.na(b); max(abs(a[ok] - b[ok]) / pmax(abs(b[ok]), 1e-300)) }
    message(sprintf("  HIP vs DESeq2: lfc %.3g  pvalue %.3g  padj %.3g  NA pattern equal: %s", rel(fit$log2FoldChange, res$log2FoldChange),
                    rel(fit$pvalue, res$pvalue), rel(fit$padj, res$padj), identical(is.na(fit$padj), is.na(res$padj))))
  }
}

## the generators behind estimateDispersionsPriorVar (d.f. <= 3): the first draws after its set.seed(2)
set.seed(2); u <- runif(3)
set.seed(2); z <- rnorm(3)
set.seed(2); e <- rexp(3)
set.seed(2); g1 <- rchisq(5, df = 1); set.seed(2); g2 <- rchisq(5, df = 2); set.seed(2); g3 <- rchisq(5, df = 3)
writeLines(c(paste("runif", paste(sprintf("%.17g", u), collapse = " ")), paste("rnorm", paste(sprintf("%.17g", z), collapse = " ")),
             paste("rexp", paste(sprintf("%.17g", e), collapse = " ")), paste("rchisq1", paste(sprintf("%.17g", g1), collapse = " ")),
             paste("rchisq2", paste(sprintf("%.17g", g2), collapse = " ")), paste("rchisq3", paste(sprintf("%.17g", g3), collapse = " "))),
           file.path(outdir, "rng_check.txt"))
