// nwk_tile_f16y_raw.hip -- the two-issuing-waves fp16 tile kernel (nwk_tile_f16y.h), raw fp16 sums (PRE = 0)
#define Y_PRE_INSTANCE 0
#include "nwk_tile_f16y.h"
