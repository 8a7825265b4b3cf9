# icikt_mi355x.R -- R-side drop-in for the reference's hot path, over icikt_rglue.c.
#
# NOT runnable in the build container (no R there).  It keeps the reference's R-level API and return
# shapes (R/kendalltau.R:96-179, R/RcppExports.R:62-64) and only swaps the places where the
# reference crosses into native code or into worker processes:
#   ici_split()  (R/kendalltau.R:280-308)  -> one .Call per pair LIST  (ici_split_gpu)
#   ici_kt()     (R/RcppExports.R:62-64)   -> one .Call for one pair   (ici_kt_gpu)
#   computation$split_fun(split_comparisons, ici_split, ...)  (R/kendalltau.R:158)
#                                          -> ONE .Call for all chunks, n_gpu GPUs inside it (ici_split_all_gpu)
#   ici_kendalltau(return_matrix = TRUE) below its argument checks (R/kendalltau.R:117-176)
#                                          -> ONE .Call for the whole matrix (ici_kendalltau_gpu): exclusion rule,
#                                             pair kernels and scale_and_reshape on the device
# With the first three, everything around them (setup_missing_matrix, setup_comparisons, scale_and_reshape) is the
# reference's own R code and stays untouched.

.icikt_warn = c(
  "2" = "Warning: The vectors only have a single value, NA returned!",
  "3" = "Warning: Either 'X' or 'Y' have only a single unique value, NA returned!",
  "4" = "Warning: Ties equal the total, NA returned!")

.icikt_count_names = c("n_entry", "missingness", "dis", "n_tie", "x_tie", "y_tie", "x0", "x1", "y0", "y1", "tot")

# Replacement for ici_split(): same arguments, same returned data.frame.  n_gpu > 1 spreads the chunk over
# devices device .. device + n_gpu - 1 inside the one call.
ici_split_gpu = function(do_comparisons, exclude_data, perspective, do_log_memory, alternative, continuity,
                         device = 0L, n_gpu = 1L) {
  storage.mode(exclude_data) = "double"          # Rcpp coerces integer input the same way
  pi = match(do_comparisons[, 1], colnames(exclude_data))
  pj = match(do_comparisons[, 2], colnames(exclude_data))
  res = .Call("icikt_R_pairs", exclude_data, as.integer(pi), as.integer(pj), perspective, alternative,
              continuity, as.integer(device), as.integer(n_gpu), FALSE)
  for (r in res$reason[res$reason > 1L]) warning(.icikt_warn[[as.character(r)]], call. = FALSE)
  do_comparisons$raw = res$raw
  do_comparisons$pvalue = res$pvalue
  do_comparisons$taumax = res$taumax
  do_comparisons$completeness = res$completeness
  do_comparisons
}

# Replacement for the dispatch line of ici_kendalltau() (R/kendalltau.R:158)
#   split_cor = computation$split_fun(split_comparisons, ici_split, exclude_data, perspective, ...)
# `split_comparisons` is the list of `core` chunks that setup_comparisons() made (consecutive blocks of
# ceiling(n_todo / ncore) pairs, R/kendalltau.R:250-255).  HIP must not be driven from forked furrr workers, so the
# chunks are bound back together (they are consecutive), computed by ONE .Call that uses all n_gpu GPUs --
# inside it the same ceiling(n_todo / n_gpu) blocks go to the GPUs, each GPU sorts 1 / n_gpu of the columns, and
# RCCL moves the prepared columns and the results over xGMI -- and split again, so that scale_and_reshape()
# (R/kendalltau.R:357-421) receives exactly what split_fun would have returned.
ici_split_all_gpu = function(split_comparisons, exclude_data, perspective, do_log_memory, alternative, continuity,
                             n_gpu = length(split_comparisons), device = 0L) {
  all_comparisons = do.call(rbind, split_comparisons)
  done = ici_split_gpu(all_comparisons, exclude_data, perspective, do_log_memory, alternative, continuity,
                       device = device, n_gpu = n_gpu)
  split(done, done$core)
}

# Replacement for ici_kt(): identical defaults and a named numeric(4); output != "simple" prints the report of
# src/kendallc.cpp:342-363 (same labels, std::to_string = "%f" formatting) from the integer counts record.
ici_kt_gpu = function(x, y, perspective = "local", alternative = "two.sided", continuity = FALSE,
                      output = "simple", device = 0L) {
  if (length(x) != length(y)) stop("'X' and 'Y' are not the same length!")
  m = cbind(as.double(x), as.double(y))
  want_report = !identical(output, "simple")
  res = .Call("icikt_R_pairs", m, 1L, 2L, perspective, alternative, continuity, as.integer(device), 1L, want_report)
  if (res$reason > 1L) warning(.icikt_warn[[as.character(res$reason)]], call. = FALSE)
  out = c(tau = res$raw, pvalue = res$pvalue, tau_max = res$taumax, completeness = res$completeness)
  if (want_report && res$reason == 0L) {   # the reference returns before its report for every NA case
    k = stats::setNames(as.numeric(res$counts[1, ]), .icikt_count_names)
    keep = if (identical(perspective, "local")) !(is.na(m[, 1]) & is.na(m[, 2])) else rep(TRUE, nrow(m))
    x2 = m[keep, 1]; y2 = m[keep, 2]
    min_x = min(x2, na.rm = TRUE) - 0.1; min_y = min(y2, na.rm = TRUE) - 0.1
    x2[is.na(x2)] = min_x; y2[is.na(y2)] = min_y
    sum_obs = sum(!duplicated(cbind(x2, y2))) + 1        # joint runs + 1 (kendallc.cpp:261-263)
    n = k[["n_entry"]]; mm = n * (n - 1)
    con_minus_dis = k[["tot"]] - k[["x_tie"]] - k[["y_tie"]] + k[["n_tie"]] - 2 * k[["dis"]]
    var = (mm * (2 * n + 5) - k[["x1"]] - k[["y1"]]) / 18 + (2 * k[["x_tie"]] * k[["y_tie"]]) / mm +
      k[["x0"]] * k[["y0"]] / (9 * mm * (n - 2))
    s_adjusted = out[["tau"]] * sqrt((mm / 2 - k[["x_tie"]]) * (mm / 2 - k[["y_tie"]]))
    if (continuity) s_adjusted = sign(s_adjusted) * (abs(s_adjusted) - 1)
    f = function(v) sprintf("%f", v); d = function(v) sprintf("%.0f", v)
    cat("min_x: ", f(min_x), "\nmin_y: ", f(min_y), "\nn_entry: ", d(n), "\nmissingness: ", d(k[["missingness"]]),
        "\ncompleteness: ", f(out[["completeness"]]), "\ntot: ", d(k[["tot"]]), "\nsum_obs: ", d(sum_obs),
        "\ndis: ", d(k[["dis"]]), "\ncon_minus_dis (k_numerator): ", f(con_minus_dis), "\nn_tie: ", f(k[["n_tie"]]),
        "\nm: ", d(mm), "\nx_tie: ", f(k[["x_tie"]]), "\ny_tie: ", f(k[["y_tie"]]), "\ns_adjusted: ", f(s_adjusted),
        "\nvar: ", f(var), "\nz_b: ", f(s_adjusted / sqrt(var)), "\ntau: ", f(out[["tau"]]),
        "\ntau_max:", f(out[["tau_max"]]), "\npvalue: ", f(out[["pvalue"]]), "\n", sep = "")
  }
  out
}

# Replacement for the whole body of ici_kendalltau() below its argument checks when return_matrix = TRUE
# (R/kendalltau.R:117-176): setup_missing_matrix, the pair list, ici_split over all chunks and scale_and_reshape run
# inside ONE .Call on the device(s) -- no masked copy of the matrix, no data.frame of n (n - 1) / 2 rows, no name-indexed
# fill of five matrices.  `data_matrix` is the RAW matrix; the result has the reference's names and shapes
# (cor, raw, pvalue, taumax, completeness: S x S with dimnames; keep: S x n_feature logical; run_time).
# include_only: handled by the reference's own setup_comparisons() (its three forms), whose pair list is handed over.
ici_kendalltau_gpu = function(data_matrix, global_na = c(NA, Inf, 0), perspective = "global", scale_max = TRUE,
                              diag_good = TRUE, include_only = NULL, alternative = "two.sided", continuity = FALSE,
                              device = 0L, n_gpu = 1L) {
  storage.mode(data_matrix) = "double"
  samples = colnames(data_matrix)
  if (is.null(samples)) { samples = paste0("s", seq_len(ncol(data_matrix))); colnames(data_matrix) = samples }
  pi = pj = NULL
  if (!is.null(include_only) || !diag_good) {   # the reference's own filter; all pairs of the triangle need no list
    cmp = do.call(rbind, setup_comparisons(samples, include_only, diag_good, 1L))
    pi = match(cmp[, 1], samples); pj = match(cmp[, 2], samples)
  }
  # the device-side exclusion rule holds up to 32 distinct finite global_na values (the reference loops over any number,
  # R/utils.R:16-20): a longer list is applied here, by the reference's own setup_missing_matrix, and NA handed over.
  # keep = t(!exclude_loc) (R/kendalltau.R:417) then comes from this mask: the masked matrix would count data NaN in.
  exclude_loc = NULL
  if (length(unique(global_na[is.finite(global_na)])) > 32L) {
    exclude_loc = setup_missing_matrix(data_matrix, global_na)
    data_matrix[exclude_loc] = NA
    global_na = NA_real_          # every excluded cell is NA now; cells that were NaN without NA in global_na are
  }                               # put right below (n_good, keep)
  t1 = Sys.time()
  res = .Call("icikt_R_matrix", data_matrix, as.double(global_na), pi, pj, perspective, alternative, continuity,
              scale_max, diag_good, as.integer(device), as.integer(n_gpu))
  t2 = Sys.time()
  if (!is.null(exclude_loc)) {    # n_good = colSums(!exclude_loc) of the CALLER's rule (R/kendalltau.R:375-386)
    res$keep = t(!exclude_loc)
    if (diag_good) {
      n_good = colSums(!exclude_loc)
      diag(res$raw) = diag(res$cor) = n_good / max(n_good)
      diag(res$completeness) = n_good / nrow(exclude_loc)
    }
  }
  for (code in 2:4) for (k in seq_len(res$reason_counts[code + 1])) warning(.icikt_warn[[as.character(code)]], call. = FALSE)
  out = res[c("cor", "raw", "pvalue", "taumax", "completeness")]
  for (nm in names(out)) dimnames(out[[nm]]) = list(samples, samples)
  out$keep = res$keep
  dimnames(out$keep) = list(samples, rownames(data_matrix))
  out$run_time = as.numeric(difftime(t2, t1, units = "secs"))
  out
}

# In ici_kendalltau() (R/kendalltau.R:127-160) the change is two lines:
#   ncore = n_gpu                       # instead of check_furrr()'s future::nbrOfWorkers(): `core` chunks = GPUs
#   split_cor = ici_split_all_gpu(split_comparisons, exclude_data, perspective, do_log_memory, alternative,
#                                 continuity, n_gpu = n_gpu)
# With n_gpu = 1 this is ici_split_gpu() on the single chunk.
