#define VSZIP_BB_T uint16_t
#define VSZIP_BB_RLO 1
#define VSZIP_BB_RHI 8
#define VSZIP_BB_FN vszip_bb_ct_u16_a
#include "boxblur_ct_inst.inc"
