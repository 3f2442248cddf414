#define VSZIP_BB_T uint8_t
#define VSZIP_BB_RLO 16
#define VSZIP_BB_RHI 22
#define VSZIP_BB_FN vszip_bb_ct_u8_c
#include "boxblur_ct_inst.inc"
