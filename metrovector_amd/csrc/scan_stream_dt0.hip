#define MVF_SCAN_DT 0
#include "scan_stream.inc"
