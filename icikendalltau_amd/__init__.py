"""icikendalltau_amd -- MI355X-native all-pairs ICI-Kendall-tau (drop-in for ICIKendallTau's
ici_kendalltau() / ici_kt() path).  See DESIGN.md and include/icikt.h."""
from ._lib import Context, IciktError, build, default_context, device_count  # noqa: F401
from .api import (HipEngine, IciKtResult, ici_kendalltau, ici_kt, ici_kt_counts, kt_fast,  # noqa: F401
                  pairwise_completeness, setup_comparisons, setup_missing_matrix)

from .formats import cor_matrix_2_long_df, long_df_2_cor_matrix, read_r_data, read_r_matrix  # noqa: F401

__version__ = "0.1.0"
