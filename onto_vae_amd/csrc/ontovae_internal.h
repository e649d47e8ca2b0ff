// Internal declarations shared by the host packer and the HIP kernels of libontovae_hip.so.
#ifndef ONTOVAE_INTERNAL_H
#define ONTOVAE_INTERNAL_H

#include "../../include/ontovae_hip.h"

#define ONTOVAE_ABI_VERSION 8

#ifdef __cplusplus
extern "C++" {
void ontovae_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
}
#endif

#endif
