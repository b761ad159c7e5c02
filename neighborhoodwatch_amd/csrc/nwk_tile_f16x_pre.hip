// nwk_tile_f16x_pre.hip -- the PRE = 1 (pre-screen) instances of the ping-pong fp16 tile kernel; see nwk_tile_f16x.h
#define X_PRE_INSTANCE 1
#include "nwk_tile_f16x.h"
