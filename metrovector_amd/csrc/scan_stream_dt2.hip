#define MVF_SCAN_DT 2
#include "scan_stream.inc"
