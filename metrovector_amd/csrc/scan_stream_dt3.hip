#define MVF_SCAN_DT 3
#include "scan_stream.inc"
