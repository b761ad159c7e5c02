// nwk_tile_f16y_pre.hip -- the two-issuing-waves fp16 tile kernel (nwk_tile_f16y.h), pre-screen instances (PRE = 1)
#define Y_PRE_INSTANCE 1
#include "nwk_tile_f16y.h"
