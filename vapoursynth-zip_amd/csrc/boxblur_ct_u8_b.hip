#define VSZIP_BB_T uint8_t
#define VSZIP_BB_RLO 9
#define VSZIP_BB_RHI 15
#define VSZIP_BB_FN vszip_bb_ct_u8_b
#include "boxblur_ct_inst.inc"
