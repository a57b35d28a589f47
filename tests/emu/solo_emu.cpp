// Host emulation build of the KERNEL SOURCE (solo_amd/csrc/*.h) -- test infrastructure only.
//
// The codec kernels are written wave-uniform (solo_wave.h); compiled without hipcc the same source
// runs with SX_NLANES == 1, which lets the CPU-side tests (-m "not gpu") check the kernel source
// against the reference / golden vectors in a container that has no GPU.  Nothing in the product
// library (solo_amd/csrc/solo_api.hip) links or calls this file.
#include <stdlib.h>
#include <string.h>
#include "../../solo_amd/csrc/solo_dec.h"

extern "C" {

struct EmuDec { SxDecState st; SxDecWork w; int useMDIndex; };

void* emu_dec_create(int useMDIndex) {
    EmuDec* d = (EmuDec*)calloc(1, sizeof(EmuDec));
    sx_dec_state_init(&d->st);
    d->useMDIndex = useMDIndex;
    return d;
}
void emu_dec_destroy(void* h) { free(h); }
int emu_dec_packet(void* h, const uint8_t* bits, int nBytes0, int nBytes1, int lostflag, int16_t* pcm) {
    EmuDec* d = (EmuDec*)h;
    return sx_decode_packet(&d->st, &d->w, bits, nBytes0, nBytes1, lostflag, d->useMDIndex, pcm);
}
int emu_sizeof_dec_state() { return (int)sizeof(SxDecState); }
int emu_sizeof_dec_work() { return (int)sizeof(SxDecWork); }

}  // extern "C"

// debug aid: raw view of the decoder state (tests only)
extern "C" const void* emu_dec_state_ptr(void* h) { return &((EmuDec*)h)->st; }
