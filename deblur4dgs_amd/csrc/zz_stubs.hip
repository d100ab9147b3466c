#include "common.h"
int d4gs_raster_bwd_impl(const D4gsDims *, const D4gsProjOut *, const D4gsIsect *, const D4gsRaster *,
                         const D4gsRasterGrads *, hipStream_t) { d4gs_set_error("not built"); return D4GS_EINVAL; }
int d4gs_project_bwd_impl(const D4gsDims *, const D4gsProjIn *, const D4gsProjOut *, const float *, const float *,
                          const float *, const float *, const float *, const D4gsLeafGrads *, hipStream_t) { d4gs_set_error("not built"); return D4GS_EINVAL; }
extern "C" size_t d4gs_bwd_partials_elems(const D4gsDims *) { return 0; }
