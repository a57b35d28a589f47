// solo_enc_k_wb.hip -- the encoder kernels compiled for the 32 kHz API rate; same source as solo_enc_k.hip.
#define SX_FS_KHZ 16
#include "solo_enc_k.hip"
