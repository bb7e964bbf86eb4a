// Int8 SHADOW rows of a Float32 / Float16 corpus (shadow_i8.hip; api.hip: scan path 6, one to four queries): K1 with
// int8 queries from the query preparation, exact i32 dot products and float keys dot * xscale[r] * qaux0[q].
#define MVF_SCAN_DT 2
#define MVF_SCAN_XS 1
#include "scan_stream.inc"
