/* SKP_Silk_typedef.h -- the scalar type names the reference's public header pulls in
 * (JC1_SDK_SRC_ARM/interface/SKP_Silk_typedef.h:42-60), so that callers written against
 * interface/AGR_JC1_SDK_API.h compile unchanged against libsolo_mi355x.so.
 * Only the names a caller of the six AGR_Sate_* entry points can need are provided. */
#ifndef SOLO_COMPAT_SKP_SILK_TYPEDEF_H
#define SOLO_COMPAT_SKP_SILK_TYPEDEF_H
#include <stdint.h>
typedef int8_t   SKP_int8;
typedef uint8_t  SKP_uint8;
typedef int16_t  SKP_int16;
typedef uint16_t SKP_uint16;
typedef int32_t  SKP_int32;
typedef uint32_t SKP_uint32;
typedef int64_t  SKP_int64;
typedef int      SKP_int;
typedef unsigned SKP_uint;
typedef int      SKP_bool;
#endif
