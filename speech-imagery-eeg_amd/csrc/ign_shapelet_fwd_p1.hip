// Forward instantiations for TT in [5,8] (split so the translation units build in parallel).
#include "ign_shapelet_fwd.h"

#define IGN_FWD_ROW(TT) \
    { { shp_fwd_launch<TT, 1, DIST_L1>, shp_fwd_launch<TT, 2, DIST_L1>, shp_fwd_launch<TT, 5, DIST_L1> }, \
      { shp_fwd_launch<TT, 1, DIST_MSE>, shp_fwd_launch<TT, 2, DIST_MSE>, shp_fwd_launch<TT, 5, DIST_MSE> }, \
      { shp_fwd_launch<TT, 1, DIST_COS>, shp_fwd_launch<TT, 2, DIST_COS>, shp_fwd_launch<TT, 5, DIST_COS> }, \
      { shp_fwd_launch<TT, 1, DIST_PEARSON>, shp_fwd_launch<TT, 2, DIST_PEARSON>, shp_fwd_launch<TT, 5, DIST_PEARSON> } }

// [TT - 5][dist][kt index: 0->1, 1->2, 2->5]
shp_fwd_launch_t ign_fwd_table_p1[4][4][3] = {
    IGN_FWD_ROW(5), IGN_FWD_ROW(6), IGN_FWD_ROW(7), IGN_FWD_ROW(8)
};
