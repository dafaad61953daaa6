// sparse.hip — FITC / VFE objectives (placeholder until the next commit).
#include "ctx.h"
int gpx_sparse_nll_grad(gpx_ctx*, int, int, const double*, int, double, double*, double*) { return GPX_ERR_UNSUPPORTED; }
int gpx_sparse_prepare(gpx_ctx*, int, int, const double*, double*) { return GPX_ERR_UNSUPPORTED; }
int gpx_sparse_predict(gpx_ctx*, const double*, long, double*, double*) { return GPX_ERR_UNSUPPORTED; }
