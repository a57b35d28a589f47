/* AGR_JC1_SDK_API.h -- drop-in name of the reference's public header
 * (JC1_SDK_SRC_ARM/interface/AGR_JC1_SDK_API.h:1-71, byte-identical in JC1_SDK_SRC_FLP/interface/).
 *
 * A caller that does  #include "AGR_JC1_SDK_API.h"  -- the reference's own test/enc_main.c:6 and
 * test/dec_main.c:6 do nothing else -- compiles unchanged with  -I<repo>/include  and links with
 * -lsolo_mi355x  instead of libJC1Codec.a: the two control structs (8 / 6 x int32, same field order)
 * and the six AGR_Sate_* prototypes live in solo_mi355x.h, this file only adds the SKP_* type names.
 * `make -C oracle dropin` + tests/test_dropin_link.py prove it on the reference's mains. */
#ifndef AGR_JC1_SDK_API_H
#define AGR_JC1_SDK_API_H
#include "SKP_Silk_typedef.h"
#include "solo_mi355x.h"
#endif
