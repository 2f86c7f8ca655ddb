/* Every entry point of include/navtex_amd.h that takes an object (handle, capture ring, group, character layer, WAV
 * file, store) or a pointer it must read, called with NULL: an error code or a no-op, never a crash.  Linked against
 * libnavtex_amd.so alone, needs no GPU (tests/test_abi.py runs it in a process of its own and holds the list of entry
 * points NOT called here -- the ones that need a device -- against the header). */
#include <stdio.h>
#include <stdlib.h>
#include "navtex_amd.h"
#define T(expr) do { printf("%-40s", #expr); fflush(stdout); long long r_ = (long long)(expr); printf(" -> %lld\n", r_); } while (0)
#define V(expr) do { printf("%-40s", #expr); fflush(stdout); expr; printf(" -> ok\n"); } while (0)
int main(void)
{
    char buf[16]; double d[4]; uint64_t u[3]; int i3[3]; int16_t iq[4] = {0};
    T(nvx_create(NULL, NULL));
    V(nvx_destroy(NULL));
    T(nvx_reset(NULL)); T(nvx_stream_reset(NULL, 0));
    T(nvx_set_trace(NULL, NULL, NULL));
    T(nvx_push_iq(NULL, 0, iq, 2)); T(nvx_push_planar(NULL, 0, iq, iq, 2));
    T(nvx_stream_set_active(NULL, 0, 1)); T(nvx_stream_stats(NULL, 0, i3, u, u + 1));
    T(nvx_flush(NULL)); T(nvx_finish(NULL)); T(nvx_stream_finish(NULL, 0)); T(nvx_poll(NULL));
    T(nvx_poll_bits(NULL, 0, 0, buf, sizeof buf));
    T(nvx_process_resident(NULL, NULL, 0, 0, 1, NULL)); T(nvx_fetch_bits(NULL)); T(nvx_bit_count(NULL, 0, 0));
    T(nvx_last_kernel_ms(NULL, 0)); V(nvx_enable_timing(NULL, 1));
    T(nvx_kernel_time_stats(NULL, 0, d, u, 0)); T(nvx_cascade_wait_stats(NULL, u, u + 1, u + 2, 0));
    T(nvx_cascade_integrity_stats(NULL, u, u + 1, u + 2, 0)); T(nvx_demod_tie_stats(NULL, u, u + 1, d));
    T(nvx_enable_debug(NULL, 1)); T(nvx_debug_cascade_state(NULL, 0, buf, 16, 0)); T(nvx_debug_advance_clock(NULL, 0, 1));
    T(nvx_debug_y3(NULL, 0, 0, d, 2)); T(nvx_debug_dphi(NULL, 0, 0, d, 2));
    T((long long)(size_t)nvx_handle_stream(NULL));
    T(nvx_decode_wav(NULL, 0, "x.wav"));
    T(nvx_capture_start(NULL, 0, 1.0, NULL)); V(nvx_capture_callback(NULL, NULL, NULL, 0, 0, NULL)); T(nvx_capture_stop(NULL));
    V(nvx_capture_stats(NULL, u, u + 1, u + 2)); T(nvx_capture_error(NULL)); T(nvx_capture_stalled(NULL, u));
    V(nvx_capture_set_stall_timeout(NULL, 1.0)); T(nvx_capture_latency(NULL, u, d, d + 1, d + 2, d + 3, 0));
    T(nvx_capture_record(NULL, NULL)); V(nvx_capture_pause(NULL, 1));
    T(nvx_group_create(NULL, 0, NULL, NULL)); V(nvx_group_destroy(NULL)); T(nvx_group_reset(NULL)); T(nvx_group_size(NULL));
    T(nvx_group_member(NULL, 0, i3, i3 + 1, i3 + 2, NULL)); T(nvx_group_member_of(NULL, 0));
    T(nvx_group_process_resident(NULL, NULL, 0, 0, 1)); T(nvx_group_fetch_bits(NULL)); T(nvx_group_push_iq(NULL, 0, iq, 2));
    T(nvx_group_flush(NULL)); T(nvx_group_finish(NULL)); T(nvx_group_poll_bits(NULL, 0, 0, buf, sizeof buf)); T(nvx_group_bit_count(NULL, 0, 0));
    V(nvx_sitor_set_trace(NULL, NULL, NULL)); V(nvx_sitor_free(NULL)); V(nvx_sitor_reset(NULL)); V(nvx_sitor_receive_bit(NULL, 'B')); V(nvx_sitor_receive_bits(NULL, "BY", 2));
    T(nvx_wav_close(NULL)); T(nvx_wav_read(NULL, buf, 1)); T(nvx_wav_write(NULL, buf, 1)); T(nvx_wav_get_length(NULL));
    T(nvx_wav_get_format(NULL)); T(nvx_wav_get_num_channels(NULL)); T(nvx_wav_get_sample_rate(NULL)); T(nvx_wav_get_sample_size(NULL));
    V(nvx_wav_set_format(NULL, 1)); V(nvx_wav_set_num_channels(NULL, 2)); V(nvx_wav_set_sample_rate(NULL, 1)); V(nvx_wav_set_sample_size(NULL, 2));
    T((long long)(size_t)nvx_wav_open(NULL, 1));
    T(nvx_sitor_encode(NULL, 1, NULL, 0));
    T(nvx_synth_host(NULL, 252000, 0, 1, iq));
    T(nvx_store_open(NULL, 0, NULL)); V(nvx_store_close(NULL)); T(nvx_store_add_message(NULL, "AB01", "x", 518));
    V(nvx_store_on_message(NULL, 0, "AB01", "x", 518)); T(nvx_store_purge(NULL, 0)); V(nvx_store_stats(NULL, u, u + 1)); V(nvx_store_set_time(NULL, 0));
    T(nvx_shim_latency(u, d, d + 1, d + 2, d + 3, 0)); T(nvx_shim_flush()); T(nvx_shim_finish()); T(nvx_shim_bits(0, buf, sizeof buf)); T(nvx_shim_stats(NULL, NULL));
    T(nvx_channelise_time_stats(d, u, 0)); V(nvx_channelise_timing(0));
    T(nvx_fsm_selftest(1, 10)); T(nvx_abi_version());
    printf("null-safety ok\n");
    return 0;
}
