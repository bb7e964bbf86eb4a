// Float16 rows of a Float32 corpus' scaled-f16 shadow (api.hip, scan path 4): K1 with the per-row scale applied.
#define MVF_SCAN_DT 1
#define MVF_SCAN_XS 1
#include "scan_stream.inc"
