#define MVF_SCAN_DT 1
#include "scan_stream.inc"
