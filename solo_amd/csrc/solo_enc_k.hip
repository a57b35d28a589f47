// solo_enc_k.hip -- the encoder's analysis / range-coding / high-band + payload kernels and the launch table of the build
// (solo_enc_kernels.h).  A translation unit of their own, so that the lanes-per-stream model of the analysis / coding kernels is a
// compile-time choice apart from the decoder's: SX_ENC_GROUP = 64 (one wavefront per stream: what runs) or 32 (TWO streams per
// wavefront, a 32-lane half each -- solo_wave.h: "wave-uniform" then means uniform within the half).  The 32-lane model is NOT
// finished: it compiles (kernel wrappers, reductions and the energy scan have their 32-lane forms) but the stages that lay four
// subframes out on the four rows of a wavefront (warped autocorrelation, shape rows, Burg) still assume 64 lanes; see DESIGN.md.
#ifndef SX_ENC_GROUP
#define SX_ENC_GROUP 64
#endif
#if SX_ENC_GROUP != 64
#define SX_GROUP SX_ENC_GROUP
#endif
#include <hip/hip_runtime.h>
#include "solo_enc_kernels.h"
#if SX_FS_KHZ == 8
extern "C" const solo_enc_ops* solo_nb_enc_ops() { return &solo_enc_ops_table; }
#else
extern "C" const solo_enc_ops* solo_wb_enc_ops() { return &solo_enc_ops_table_wb; }
#endif
