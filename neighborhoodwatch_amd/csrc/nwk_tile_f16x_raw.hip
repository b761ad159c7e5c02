// nwk_tile_f16x_raw.hip -- the PRE = 0 (raw fp16 sums) instances of the ping-pong fp16 tile kernel; see nwk_tile_f16x.h
#define X_PRE_INSTANCE 0
#include "nwk_tile_f16x.h"
