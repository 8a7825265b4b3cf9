# icikt_mi355x.R -- R-side drop-in for the reference's hot path, over icikt_rglue.c.
#
# NOT runnable in the build container (no R there).  It keeps the reference's R-level API and return
# shapes (R/kendalltau.R:96-179, R/RcppExports.R:62-64) and only swaps the two places where the
# reference crosses into native code:
#   ici_split()  (R/kendalltau.R:280-308)  -> one .Call per pair LIST  (ici_split_gpu)
#   ici_kt()     (R/RcppExports.R:62-64)   -> one .Call for one pair   (ici_kt_gpu)
# Everything around them (setup_missing_matrix, setup_comparisons, scale_and_reshape) is the
# reference's own R code and stays untouched.

.icikt_warn = c(
  "2" = "Warning: The vectors only have a single value, NA returned!",
  "3" = "Warning: Either 'X' or 'Y' have only a single unique value, NA returned!",
  "4" = "Warning: Ties equal the total, NA returned!")

# Replacement for ici_split(): same arguments, same returned data.frame.
ici_split_gpu = function(do_comparisons, exclude_data, perspective, do_log_memory, alternative, continuity,
                         device = 0L) {
  storage.mode(exclude_data) = "double"          # Rcpp coerces integer input the same way
  pi = match(do_comparisons[, 1], colnames(exclude_data))
  pj = match(do_comparisons[, 2], colnames(exclude_data))
  res = .Call("icikt_R_pairs", exclude_data, as.integer(pi), as.integer(pj), perspective, alternative,
              continuity, as.integer(device))
  for (r in res$reason[res$reason > 1L]) warning(.icikt_warn[[as.character(r)]], call. = FALSE)
  do_comparisons$raw = res$raw
  do_comparisons$pvalue = res$pvalue
  do_comparisons$taumax = res$taumax
  do_comparisons$completeness = res$completeness
  do_comparisons
}

# Replacement for ici_kt(): identical defaults and a named numeric(4).
ici_kt_gpu = function(x, y, perspective = "local", alternative = "two.sided", continuity = FALSE,
                      output = "simple", device = 0L) {
  if (length(x) != length(y)) stop("'X' and 'Y' are not the same length!")
  m = cbind(as.double(x), as.double(y))
  res = .Call("icikt_R_pairs", m, 1L, 2L, perspective, alternative, continuity, as.integer(device))
  if (res$reason > 1L) warning(.icikt_warn[[as.character(res$reason)]], call. = FALSE)
  c(tau = res$raw, pvalue = res$pvalue, tau_max = res$taumax, completeness = res$completeness)
}

# In ici_kendalltau() (R/kendalltau.R:158) the only change is the dispatch line: HIP must not be driven
# from forked `multicore` workers, so the GPU path maps over the chunks in the calling process:
#   split_cor = purrr::map(split_comparisons, ici_split_gpu, exclude_data, perspective, do_log_memory,
#                          alternative, continuity)
# (one chunk per GPU when several are used: pass device = chunk index - 1).
