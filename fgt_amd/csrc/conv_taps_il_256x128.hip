// The 256 x 128 tile of the interleaved-request tap kernel as its own translation unit (build time; see the end of conv_taps_il.hip).
#define FGT_IL_PART 1
#include "conv_taps_il.hip"
