/* nvx_internal.h -- declarations shared between the translation units of
 * libnavtex_amd.so; not part of the public ABI.                              */
#ifndef NVX_INTERNAL_H
#define NVX_INTERNAL_H

#include "navtex_amd.h"
#include "nvx_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

void nvx_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

void nvx_synth_periods(const nvx_carrier *c, uint32_t sample_rate, uint64_t first, size_t count,
                       nvx_period *out);

#ifdef __cplusplus
}
#endif
#endif
