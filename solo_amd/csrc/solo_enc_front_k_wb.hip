// the persistent schedule's front kernel at the 32 kHz API rate (solo_enc_front_k.hip)
#define SX_FS_KHZ 16
#include "solo_enc_front_k.hip"
