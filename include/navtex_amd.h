/* navtex_amd.h -- C ABI of libnavtex_amd.so
 *
 * MI355X-native NAVTEX demodulation hot path: int16 IQ in, 'B'/'Y' bits and
 * SITOR-B characters out.  The decimating FIR cascade, the FSK discriminator
 * and the bit-timing / mark-space matched filters run as HIP kernels on
 * gfx950; the SITOR-B character layer runs on the host in C.
 *
 * Plain pointers and sizes only.  Every entry point cites the interface of the
 * reference receiver (bartelvdh/Navtex, paths relative to its repo root) that
 * it replaces or stands behind.  There is NO CPU fallback: device entry points
 * return NVX_ERR_NODEV / NVX_ERR_HIP when no gfx950 device is usable.
 */
#ifndef NAVTEX_AMD_H
#define NAVTEX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVX_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ errors */
enum {
    NVX_OK          =  0,
    NVX_ERR_ARG     = -1,   /* bad argument / size not a multiple of the frame   */
    NVX_ERR_NODEV   = -2,   /* no HIP device (this library has no CPU path)      */
    NVX_ERR_HIP     = -3,   /* a HIP runtime call failed; see nvx_last_error()   */
    NVX_ERR_NOMEM   = -4,
    NVX_ERR_STATE   = -5,   /* call not valid in the handle's current mode       */
    NVX_ERR_IO      = -6    /* file errors of the WAV path                       */
    /* (-7 was NVX_ERR_FULL of ABI 1, "staging full": never returned since the streams of a handle launch
     * independently; the value stays unused so that no later code is mistaken for it)                    */
};
NVX_API const char *nvx_last_error(void);
/* "navtex_amd <abi>.<minor> (gfx950)".  NVX_ABI_VERSION counts incompatible changes of this header (struct layouts,
 * signatures): 2 = nvx_config begins with struct_size, nvx_capture_error lost its second parameter, NVX_ERR_FULL is
 * gone.  nvx_abi_version() returns the library's; a caller built against another value must be rebuilt.          */
#define NVX_ABI_VERSION 2
NVX_API const char *nvx_version(void);
NVX_API int nvx_abi_version(void);

/* ------------------------------------------------------------- rate algebra */
#define NVX_RATE_RAW      2016000   /* receiver/capt_sched.c:31-34 H_SAMPLE_RATE          */
#define NVX_RATE_IN        252000   /* receiver/capt_sched.c:29   IN_SAMPLE_RATE          */
#define NVX_DECIM0              8   /* receiver/capt_sched.c:31   H_DECIMATION_FACTOR     */
#define NVX_SPB_IN           2520   /* 252 kS/s samples per 100-baud bit                  */
#define NVX_FRAME_BITS         32   /* device launch granule: 32 bit periods = 0.32 s     */
#define NVX_FRAME_IN   (NVX_SPB_IN * NVX_FRAME_BITS)      /* 80 640 samples @252 kS/s     */
#define NVX_FRAME_RAW  (NVX_FRAME_IN * NVX_DECIM0)        /* 645 120 samples @2.016 MS/s  */
#define NVX_FRAME_Y3   (NVX_FRAME_BITS * 9)               /* 288 samples @900 S/s / chain */

/* chain_mask bits: which of the two +-14 kHz chains of a stream are decoded
 * (receiver/nav_sched.C:10-17 wires both; receiver/fir2cpp.C:112-128)       */
#define NVX_CHAIN_518 1u    /* carrier at +14 kHz of the stream centre, mixed down */
#define NVX_CHAIN_490 2u    /* carrier at -14 kHz of the stream centre, mixed up   */

/* ==========================================================================
 * A. Reference-compatible push surface.
 *    Replaces receiver/fir1cpp.o fir2cpp.o fir3cpp.o decoder.o nav_b_sm.o
 *    nav_sched.o at link time: an unmodified receiver/capt_sched.c calls these
 *    three symbols (declarations capt_sched.c:17-19; calls :511, :554, :612).
 *    The contract as capt_sched.c uses it: no return value, process-global
 *    singleton, one caller thread, each init called once before the first
 *    sample, int16 values cast to double.  Samples are buffered into frames and
 *    run on GPU 0; characters leave through add_message().  On the same samples
 *    the bits and messages are the reference's.
 *
 *    Where this surface deliberately does NOT behave like the reference
 *    (DESIGN.md section 4.4 has the table, tests/test_deviations.py the tests):
 *    1. Re-initialisation.  The reference's init_fir_filter1() clears FIR1's
 *       ring and counter only (receiver/fir1cpp.C:65-77); its init_fir2_wrapper()
 *       clears the 518 chain's FIR2 ring and the mixer index both chains share
 *       (receiver/fir2cpp.C:90-110) -- never the 490 chain's FIR2 statics, FIR3,
 *       the decoders or the character layers.  Called once at start-up, as
 *       capt_sched.c does (:552-555, :612), that is "everything zero".  Called
 *       AGAIN in mid-stream the reference goes on with half its pipeline
 *       cleared; this library cannot represent that state (a frame is the span
 *       after which all decimation counters are at phase 0 together) and does
 *       not try: here init_fir_filter1() always starts a NEW stream -- all
 *       carried state zeroed, samples buffered but not yet decoded dropped --
 *       and a repeated init_fir2_wrapper() changes nothing.
 *    2. Input domain.  The reference filters any double; sample_in_1 here takes
 *       int16 VALUES (what capt_sched.c:511 passes).  Anything else -- a
 *       fraction, a value beyond the int16 range, a NaN -- is rounded to the
 *       nearest int16 (ties to even, NaN to 0) and counted: nvx_shim_stats.
 *       nvx_sample_to_int16 is that conversion (returns 1 when v was an int16
 *       value already).
 *    3. Long runs.  The reference's decoder counts samples in an `int`
 *       (receiver/decoder.h:60, decoder.C:75,85) that passes INT_MAX after
 *       2^31 samples at 900 S/s = 27.6 days; from then on its bit timing test
 *       (bd_seq_nbr % 9 == offset) fails for every offset but 0 and it falls
 *       nearly silent for the next 27.6 days.  The bit phase here is a function
 *       of the sample's position; decoding simply goes on.
 *    4. A sample behind nvx_shim_finish starts a new stream (the reference
 *       never ends one).
 * ========================================================================== */
NVX_API void init_fir_filter1(void);                       /* receiver/fir1cpp.h:2  */
NVX_API void sample_in_1(double sample_I, double sample_Q);/* receiver/fir1cpp.h:3  */
NVX_API void init_fir2_wrapper(void);                      /* receiver/nav_sched.h:1 */
NVX_API int  nvx_sample_to_int16(double v, int16_t *out);
/* sample_in_1 calls since the library was loaded, and how many of them carried a value outside the input domain
 * (the first one also prints a line to stderr and sets nvx_last_error); either pointer may be NULL              */
NVX_API int  nvx_shim_stats(uint64_t *samples, uint64_t *off_domain);

/* Characters-out sink, receiver/message_store.h:7 (impl message_store.c:59-97).
 * The library only CALLS this symbol.  A weak default that prints the message
 * is provided so the library links stand-alone; the receiver's own
 * message_store.o overrides it.                                             */
int add_message(char *bbbb, char *message, int freq);

/* Drains whatever the push surface has buffered so far in whole frames and
 * waits for the GPU (the reference has no equivalent: it never terminates).
 * Results of launched frames reach add_message without it: a few ms after the
 * sample that completed their frame, through the singleton's housekeeping
 * thread (nvx_poll every 2 ms while a launch is in flight, every 50 ms
 * otherwise; nvx_shim_latency reports the figure).  What only this call delivers is the tail of a capture that
 * STOPS: samples sit in the singleton's 4096-sample buffer and in frames not
 * yet complete (up to 0.32 s of signal); a program that ends a capture calls
 * it once, an endless receiver (capt_sched.c:618-621) never needs to.       */
NVX_API int nvx_shim_flush(void);
/* END of the input (a file replayed through sample_in_1, a capture that is over): nvx_shim_flush, then the last,
 * partial frame at its true length -- nvx_finish of section C on the singleton.  Afterwards the singleton has decoded
 * exactly the bits the reference's objects have decoded when capt_sched.c's loop (:509-513) has handed them the same
 * samples and stopped.  The next sample (or init_fir_filter1) starts a new stream.                                   */
NVX_API int nvx_shim_finish(void);
/* Decode latency of this surface, per frame of 0.32 s: from the entry of the sample_in_1 call (or singleton stream
 * callback) that carried the frame's last sample to its bits being pollable and its messages delivered to
 * add_message -- booked as nvx_capture_latency books a capture ring's (section B').  The reference decodes inline
 * (receiver/nav_b_sm.C:87 is called from inside sample_in_1's call chain); a character here is at most 0.32 s (its frame
 * still filling) + this behind it.  frames: latencies booked since init_fir_filter1; p50 / p99 over the last 8192, max
 * and last in ms (-1: none yet); reset != 0 clears afterwards.                                                       */
NVX_API int nvx_shim_latency(uint64_t *frames, double *p50_ms, double *p99_ms, double *max_ms, double *last_ms, int reset);
/* Bits the singleton has produced so far for chain 0 (518) / 1 (490).        */
NVX_API size_t nvx_shim_bits(int chain, char *out, size_t cap);

/* ==========================================================================
 * B. SDRplay stream-callback shape, receiver/capt_sched.c:105
 *      void StreamACallback(short *xi, short *xq, sdrplay_api_StreamCbParamsT
 *           *params, unsigned int numSamples, unsigned int reset, void *ctx)
 *    Planar int16 arrays owned by the caller, valid only during the call.
 *    `params` is opaque here (sdrplay_api.h is vendor-proprietary), `reset`
 *    and `params` are ignored exactly as the reference ignores them.
 *    cbContext must be an nvx_handle* in push mode, or NULL for the singleton
 *    of section A (stream 0).  Thread-safe against itself (the reference
 *    serialises with a mutex, capt_sched.c:111,144).
 * ========================================================================== */
NVX_API void nvx_StreamACallback(short *xi, short *xq, void *params,
                                 unsigned int numSamples, unsigned int reset, void *cbContext);

/* ==========================================================================
 * B'. Live-capture ring: the producer/consumer structure of the reference's
 *    capture thread (receiver/capt_sched.c:105-148 producer, :443-446 ring,
 *    :484-528 consumer loop) inside the library, for callers that must never
 *    block in the vendor callback.
 *      producer  nvx_capture_callback(): StreamACallback's shape, cbContext =
 *                the nvx_capture*; copies xi/xq into an interleaved int16 ring
 *                and returns.  If the ring has no room the excess samples are
 *                DROPPED and counted (the reference's ring silently overwrites
 *                unread data instead, capt_sched.c:120-129).
 *      consumer  a library thread that wakes on new data (the reference polls
 *                every 50 ms), splits wrapped spans exactly like :494-503 and
 *                feeds nvx_push_iq().
 * ========================================================================== */
typedef struct nvx_capture nvx_capture;
/* h must be a push-mode handle; the forward declaration of nvx_handle is below */
struct nvx_handle;
NVX_API int  nvx_capture_start(struct nvx_handle *h, int stream, double ring_seconds, nvx_capture **out);
NVX_API void nvx_capture_callback(short *xi, short *xq, void *params, unsigned int numSamples,
                                  unsigned int reset, void *cbContext);
/* drains the ring, stops the consumer thread, flushes the handle */
NVX_API int  nvx_capture_stop(nvx_capture *c);
/* complex samples offered by the producer / dropped on overrun / handed to the GPU pipeline */
NVX_API void nvx_capture_stats(nvx_capture *c, uint64_t *received, uint64_t *dropped, uint64_t *consumed);
/* Health of the consumer while it runs: the error that stopped it (NVX_OK while it is alive; a failed launch or HIP
 * call is the only thing that stops it early).  The streams of a handle advance independently (section C, "stream
 * independence"), so another stream's stall produces no back-pressure.                                              */
NVX_API int  nvx_capture_error(nvx_capture *c);
/* Silent-radio detection.  The reference has one radio and only prints sdrplay_api_DeviceRemoved when it disappears
 * (receiver/capt_sched.c:210-212).  With several radios on one handle a silent one must not hold the others: when the
 * consumer has handed on no sample for the stall timeout (default 2 s; <= 0 disables), it marks its stream INACTIVE
 * (nvx_stream_set_active) -- launches stop waiting for it -- and counts the event.  nvx_capture_stalled returns 1
 * while the stream is marked silent, 0 otherwise (negative: error); the first sample that arrives afterwards makes
 * the stream active again and it continues bit-exactly from its own carried state.  The ring's timeout also becomes
 * its stream's push-level timeout (nvx_config.stall_timeout_ms), so one figure governs how long the others wait.  */
NVX_API int  nvx_capture_stalled(nvx_capture *c, uint64_t *stall_events);
NVX_API void nvx_capture_set_stall_timeout(nvx_capture *c, double seconds);
/* Decode latency of the live path.  The reference decodes synchronously per sample and calls add_message inline
 * (receiver/capt_sched.c:484-528 with its 50 ms poll, receiver/nav_b_sm.C:87); here a frame (0.32 s) waits for its last
 * sample, a launch and a collect.  The ring stamps the moment the callback that carried a frame's LAST sample was
 * entered; when the launch covering the frame has been collected (bits pollable, messages delivered) the latency
 * collect - arrival is booked.  frames: latencies booked so far; p50 / p99 over the last 8192 of them, max and last in
 * ms (-1: none yet); reset != 0 clears afterwards.  The bound a receiver can rely on for a character: 0.32 s (its frame
 * still filling) + what this call reports (launch + collect, a few ms; 50 ms at worst when no callback wakes the
 * consumer) -- INTEGRATION.md section 1.  nvx_reset of the handle ends the bookkeeping of an attached ring (the
 * frames no longer line up); the figures booked so far stay readable.                                                */
NVX_API int  nvx_capture_latency(nvx_capture *c, uint64_t *frames, double *p50_ms, double *p99_ms, double *max_ms, double *last_ms, int reset);
/* debug recording (the reference's debug_mode, capt_sched.c:87-101 PrepWav/EndWav and :516):
 * every span the consumer hands to the pipeline is also appended to a 2-channel 16-bit WAV
 * at the handle's input rate.  filename NULL stops and closes; nvx_capture_stop closes too. */
NVX_API int  nvx_capture_record(nvx_capture *c, const char *filename);
/* test hook: pause (1) / resume (0) the consumer, to provoke an overrun deterministically */
NVX_API void nvx_capture_pause(nvx_capture *c, int paused);

/* ==========================================================================
 * C. Block API (what A and B adapt onto).
 * ========================================================================== */
typedef struct nvx_handle nvx_handle;

/* message sink; same argument meaning as add_message, plus the stream index */
typedef void (*nvx_message_fn)(void *user, int stream, const char *bbbb, const char *message, int freq);

typedef struct nvx_config {
    uint32_t struct_size;     /* sizeof(nvx_config) of the header the CALLER was built with: set by      */
                              /* nvx_config_default, checked by nvx_create / nvx_group_create -- a caller */
                              /* built against another layout gets NVX_ERR_ARG instead of a struct read   */
                              /* past its end.  First member, so that it sits inside every version.       */
    int      device;          /* HIP device ordinal                                        */
    int      n_streams;       /* independent IQ streams on this GPU                        */
    int      raw_rate;        /* 1: input at 2.016 MS/s, integer stage 0 (/8) on device;   */
                              /* 0: input at 252 kS/s exactly as StreamACallback delivers  */
    uint32_t chain_mask;      /* default for every stream (NVX_CHAIN_518 | NVX_CHAIN_490)  */
    const uint8_t *chain_masks; /* optional per-stream override, n_streams entries         */
    const int *labels;        /* optional [n_streams][2] freq labels (default 518, 490)    */
    int      max_frames;      /* largest number of frames one launch will be given         */
    int      char_layer;      /* 1: run the host SITOR-B layer on the decoded bits         */
    nvx_message_fn on_message;/* NULL: call add_message(bbbb, message, freq)               */
    void    *user;
    int      push_mode;       /* 1: allocate pinned staging for nvx_push_* (host input)    */
    int      wideband;        /* 1: n_streams counts WIDEBAND inputs at 2.016 MS/s (raw_rate is  */
                              /* ignored); each is channelised (section G) into 8 sub-bands that  */
                              /* run through the 252 kS/s path.  Decoded stream index = 8*w + k   */
                              /* (k = sub-band, centre k*252 kHz), so chain_masks / labels have   */
                              /* 8*n_streams entries.                                             */
    int      bit_history;     /* decoded bits kept per chain for nvx_poll_bits (0 = NVX_BIT_HISTORY) */
    int      host_threads;    /* threads the character layer of one collect may use (0 = NVX_HOST_THREADS or   */
                              /* the hardware's count, at most 16)                                             */
    int      stage0_order;    /* raw_rate = 1 only.  0 / 1: stage 0 is an integrate-and-dump over 8 samples,          */
                              /*   out = (sum + 4) >> 3 (25 dB of alias rejection at the NAVTEX offsets);             */
                              /* 3: three such boxcars in cascade (a third-order CIC as its 22-tap FIR,               */
                              /*   w = 1 3 6 10 15 21 28 36 42 46 48 48 46 ... 3 1):  out = (sum w x + 256) >> 9,     */
                              /*   76 dB of alias rejection, 14 samples of carried history, ~2 % more kernel time.    */
                              /* Both are build-owned integer definitions (the reference starts at 252 kS/s:          */
                              /* receiver/capt_sched.c:31-34); anything else is NVX_ERR_ARG.                           */
    int      eager_launch;    /* push_mode = 1 only.  0: a launch goes out when EVERY active stream has a whole frame  */
                              /*   (fewest, largest launches: thousands of host-fed streams).  1: as soon as ANY        */
                              /*   stream has one, with the streams that have -- for a handful of free-running radios,  */
                              /*   whose frames complete at different moments: nobody's bits wait for the slowest       */
                              /*   radio's frame (up to 0.32 s otherwise).  The streams are independent receivers       */
                              /*   either way (receiver/decoder.h:31-60, receiver/nav_b_sm.h:92-114): same bits.        */
    int      stall_timeout_ms;/* push_mode = 1 only.  A stream that has delivered nothing for this long is not waited  */
                              /*   for by the others' launches (0: the default, 2000; < 0: waited for for ever).  Its   */
                              /*   next push counts again at once.                                                     */
} nvx_config;

/* Callbacks (on_message) run on the thread that calls nvx_flush / nvx_fetch_bits / nvx_push_* with the
 * handle locked: they must not call back into the same handle.                                      */
NVX_API void nvx_config_default(nvx_config *cfg);
NVX_API int  nvx_create(const nvx_config *cfg, nvx_handle **out);
NVX_API void nvx_destroy(nvx_handle *h);
/* Trace sink of the handle's character layers: receives the text the reference prints to stdout from its character
 * layer ("phasing detected", "START OF MESSAGE", "line added: ...", "END OF MESSAGE", ...: receiver/nav_b_sm.C), chain by
 * chain as launches are collected, on the collecting thread (for launches of many chains: on the character layer's
 * worker threads, concurrently).  NULL turns it off.  The singleton of section A prints it to stdout when
 * NAVTEX_AMD_TRACE=1.                                                                                                 */
typedef void (*nvx_trace_fn)(void *user, const char *text);
NVX_API int  nvx_set_trace(nvx_handle *h, nvx_trace_fn fn, void *user);
/* zero all carried DSP state (FIR histories, demodulator, character layer)  */
NVX_API int  nvx_reset(nvx_handle *h);
/* ... the same for ONE stream, while the others keep everything they carry: an ended stream (nvx_finish /
 * nvx_decode_wav) starts a new input -- the next file, the next capture.  Waits for the handle's launched work and
 * delivers its bits and messages first; the stream's bit counters restart at 0.  A handle whose launch failed
 * (NVX_ERR_STATE everywhere) needs nvx_reset.                                                                    */
NVX_API int  nvx_stream_reset(nvx_handle *h, int stream);

/* ---- host-input path (pinned staging + hipMemcpyAsync) -------------------
 * Interleaved I,Q int16 (the layout of the reference's sample_buffer,
 * capt_sched.c:120-129) or planar xi/xq (the callback's layout).  Samples are
 * at the handle's input rate.
 * Stream independence: the chains share no state (receiver/decoder.h:31-60,
 * receiver/nav_b_sm.h:92-114), so the streams of a handle need no common
 * clock.  A launch goes out when every ACTIVE stream has a whole frame staged
 * and covers all of them; when one stream's staging is full (max_frames + 1
 * frames) before that, the launch goes out with the streams that HAVE a frame,
 * each carrying its own filter / demodulator state -- a stalled or slower
 * radio never blocks the others, and rejoins later bit-exactly.  A stream
 * that has delivered nothing for 2 s (nvx_config.stall_timeout_ms; no capture
 * ring needed: the handle keeps the time of every stream's last push) is not
 * waited for either, so the others keep launching frame by frame; its next
 * push counts again at once.
 * Threads: any number of threads may push into one handle, ONE per stream at
 * a time (a second pusher of the same stream waits until the first one's call
 * has returned: a push call is atomic against other pushes of its stream).  Pushes of 128 KB and
 * more copy into the pinned staging without the handle's lock, so the capture
 * or replay threads of several radios fill their streams side by side; a
 * launch first waits for the copies in flight to be committed.              */
NVX_API int nvx_push_iq(nvx_handle *h, int stream, const int16_t *iq_interleaved, size_t n);
NVX_API int nvx_push_planar(nvx_handle *h, int stream, const int16_t *xi, const int16_t *xq, size_t n);
/* Mark a stream of a push-mode handle inactive (0): launches no longer wait for it (a silent radio).  Pushing to it
 * makes it active again.  nvx_stream_stats: the flag, the frames the stream has been through since create / reset,
 * and how many launches of the handle covered only some of its streams (any out pointer may be NULL).           */
NVX_API int nvx_stream_set_active(nvx_handle *h, int stream, int active);
NVX_API int nvx_stream_stats(nvx_handle *h, int stream, int *active, uint64_t *frames_done, uint64_t *partial_launches);
/* launch whatever is staged in WHOLE frames, wait for all launched work, deliver bits/messages.  A partial frame stays
 * staged: the stream goes on bit-exactly with the next push.                                                        */
NVX_API int nvx_flush(nvx_handle *h);
/* END of the input.  The reference's loop hands every sample it is given to sample_in_1 and stops
 * (receiver/capt_sched.c:509-513): its decoder has then seen floor(n / 280) samples at 900 S/s and has decided exactly
 * the bits those samples decide (receiver/decoder.C:73-137).  nvx_finish does the same for every stream of a push-mode
 * handle: nvx_flush, then ONE more launch for the streams that still hold a partial frame, each at its TRUE length --
 * the demodulator stops at the stream's last real 900 S/s sample (floor(n / 280) at 252 kS/s input, floor(n / 2240) at
 * 2.016 MS/s); no padding is decoded, no bit is withheld: the bits of a stream are then exactly the reference's on the
 * same samples, whatever the length.  Every stream that has had input is ENDED afterwards, whatever its length (with a
 * partial frame its filters have run past its last sample; on a frame boundary nothing was left to launch -- one rule, so
 * that nothing depends on a length modulo the frame): nvx_push_* and launches that name it return NVX_ERR_STATE until
 * nvx_reset / nvx_stream_reset, and the others' launches do not wait for it.  A stream that has had no input at all has no
 * end and is left as it is.  nvx_stream_finish: the same for one stream (the whole frames of the others are launched as
 * by nvx_flush).
 * Against pushes on other threads a finish is atomic per push CALL: the calls in progress on the streams that are ending
 * run to their end first and are decoded; a call that arrives meanwhile waits and is then refused whole (NVX_ERR_STATE,
 * nothing staged).  The same holds for nvx_stream_reset and nvx_reset, after which the waiting call starts the new stream. */
NVX_API int nvx_finish(nvx_handle *h);
NVX_API int nvx_stream_finish(nvx_handle *h, int stream);
/* Take in whatever launched work has ALREADY finished -- bits appended, character layer run, messages delivered on
 * the calling thread -- and return at once; never waits for the GPU.  Results otherwise reach the host with the next
 * push (every nvx_push_* takes in finished work on its way out), launch, flush or fetch; a caller with a loop of its own
 * (the capture ring's consumer and the housekeeping thread of section A's singleton do this on every wake: 2 ms apart
 * while a launch is in flight, 50 ms otherwise) calls this so that the last message before a quiet spell is not held back
 * until signal arrives again.                                                                                       */
NVX_API int nvx_poll(nvx_handle *h);
/* copy out and consume decoded bits ('B'/'Y') of one chain; returns count.
 * The receiver runs unattended for weeks (main(), receiver/capt_sched.c:558, and its
 * endless capture loop :618-621), so the library keeps
 * only the most recent cfg.bit_history bits per chain (up to twice that between
 * trims): a reader further behind resumes at the oldest bit still held.  The
 * character layer sees every bit regardless.                                 */
#define NVX_BIT_HISTORY 65536
NVX_API size_t nvx_poll_bits(nvx_handle *h, int stream, int chain, char *out, size_t cap);

/* ---- device-resident path (roofline runs; IQ already in HBM) --------------
 * d_iq: device pointer, int16 I,Q interleaved, layout [n_streams][pitch] in
 * complex samples; stream s, frame f starts at d_iq + 4*(s*pitch + f*FRAME).
 * Processes `n_frames` frames starting at frame `first_frame` of every stream
 * on `hip_stream` (a hipStream_t; NULL = the handle's own stream), carrying
 * state from the previous call.  Successive calls may name different streams:
 * the library orders a launch behind its predecessor (the carried state makes
 * launches of one handle sequential by nature).  Asynchronous; bits stay on
 * the device until nvx_fetch_bits.
 * Before anything is launched the span the kernels will read -- up to the last
 * stream's last frame -- is held against the allocation d_iq lies in
 * (hipMemGetAddressRange): a launch that would read past its end is refused
 * with NVX_ERR_ARG instead of faulting on the device.  nvx_synth_device and
 * nvx_channelise_resident check their operands the same way.                 */
NVX_API int nvx_process_resident(nvx_handle *h, const void *d_iq, size_t pitch_samples,
                                 size_t first_frame, int n_frames, void *hip_stream);
/* synchronise, run the character layer (if enabled) on the new bits          */
NVX_API int nvx_fetch_bits(nvx_handle *h);
/* total bits decoded so far on (stream, chain) since create/reset            */
NVX_API size_t nvx_bit_count(nvx_handle *h, int stream, int chain);

/* ---- instrumentation -------------------------------------------------------
 * HIP-event durations (ms) of the kernels of the most recent
 * nvx_process_resident / push launch, measured on the launch stream:
 * which = 0 FIR cascade (wideband handles: the fused channeliser + cascade
 * kernel), 1 demodulator, 2 FIR3 as its own kernel (wideband handles, whose
 * fused kernel ends at FIR2; 0 otherwise).  Valid after a synchronise.       */
NVX_API float nvx_last_kernel_ms(nvx_handle *h, int which);
NVX_API void  nvx_enable_timing(nvx_handle *h, int enabled);
/* sum of the event durations (ms) and number of timed launches collected since
 * the last reset of the statistics; reset != 0 clears them afterwards          */
NVX_API int   nvx_kernel_time_stats(nvx_handle *h, int which, double *sum_ms, uint64_t *launches, int reset);
/* hand-over statistics of the FIR-cascade work queue over the collected launches: how many units had to
 * wait for the previous frame of their stream, and how many polls (about 1 us each) they spent waiting  */
NVX_API int   nvx_cascade_wait_stats(nvx_handle *h, uint64_t *polls, uint64_t *units_waited, uint64_t *launches, int reset);
/* Integrity of the carried filter state (the reference keeps it in statics: receiver/fir1cpp.C:51-60,
 * receiver/fir2cpp.C:74-83, receiver/fir3cpp.h:90-95; here it travels between work units through HBM).  Every state
 * block carries a 64-bit word over its contents and its position; a unit recomputes it over what it loaded.
 * stale_repaired: hand-overs INSIDE a launch whose block failed the check -- the unit rebuilt its histories from its
 * own input instead (bit-identical results) -- expected 0; launch_failures: launches whose inherited state failed the
 * check.  Such a failure STICKS: the call that collects the launch returns NVX_ERR_HIP, its bits and those of every
 * launch queued behind it are discarded (what the device carries from there on was computed from the bad block), and
 * every later push, launch, flush, poll and fetch returns NVX_ERR_STATE until nvx_reset;
 * launches: launches collected so far (as nvx_cascade_wait_stats).  reset != 0 clears the two counters afterwards. */
NVX_API int   nvx_cascade_integrity_stats(nvx_handle *h, uint64_t *stale_repaired, uint64_t *launch_failures, uint64_t *launches, int reset);
/* How close the demodulator's bit-timing decisions came to a tie since create / reset.  The arg-max over the nine
 * class sums (receiver/decoder.C:202-215, strict '>') is the one decision of the path that rests on delta-phi values
 * which may differ from glibc's atan2 in the last bit; such a difference can only matter where the best sum and the
 * runner-up are closer than ~1e-15 relative.  near_ties counts timing evaluations with best > 0 and a margin below
 * 2^-40 relative (that is 4000 times wider than the last bit reaches); evaluations counts those with best > 0;
 * min_margin is the smallest (best - runner_up) / best seen (-1 when there was no evaluation).  Synchronises with
 * the handle's last launch.                                                                                      */
NVX_API int   nvx_demod_tie_stats(nvx_handle *h, uint64_t *near_ties, uint64_t *evaluations, double *min_margin);
/* self-test of the demodulator's bit-period transition table against the per-sample rule it is
 * generated from (receiver/decoder.C:62-137, 202-249), on `periods` pseudo-random bit periods;
 * returns the number of differences (0 = pass).  Host only, no device needed.                  */
NVX_API int   nvx_fsm_selftest(uint32_t seed, int periods);
/* test / diagnostics hook: move a stream's sample clock FORWARD by `periods` x 163 296 samples at 900 S/s (567 frames =
 * lcm of the frame, 288, and of the bit-timing filter's ring algebra, 9 x 567) without touching a sample of its state:
 * every index the kernels derive from the clock is the same afterwards, so decoding goes on exactly as if nothing had
 * happened -- unless some part of the path does not carry the clock in 64 bits.  tests/test_gpu_deviations.py uses it to
 * put a stream just below 2^31 samples, where the reference's `int bd_seq_nbr` overflows after 27.6 days
 * (receiver/decoder.h:60, decoder.C:75,85), and to walk it across.  Waits for the handle's launches; re-tags the seal of
 * the stream's carried FIR state for its new position.  Not for wideband handles, and not before the stream has been
 * through three frames (the priming thresholds of the timing filter are the one thing that is not periodic): NVX_ERR_STATE. */
#define NVX_CLOCK_PERIOD 163296
NVX_API int   nvx_debug_advance_clock(nvx_handle *h, int stream, uint64_t periods);
/* allocate (1) / release (0) the delta-phi debug buffer used by nvx_debug_dphi */
NVX_API int   nvx_enable_debug(nvx_handle *h, int enabled);
/* test / diagnostics hook: the carried FIR state block of one decoded stream -- the block the stream's NEXT launch will
 * read: filter histories as fp64 pairs, then the seal (nvx_cascade_integrity_stats) -- copied to host memory (write = 0)
 * or replaced from it (write = 1); `bytes` must be NVX_STATE_BLOCK_BYTES.  Synchronises with the handle's launches.     */
#define NVX_STATE_BLOCK_BYTES 4352
NVX_API int    nvx_debug_cascade_state(nvx_handle *h, int stream, void *buf, size_t bytes, int write);
/* debug tap: copy the 900 S/s FIR-cascade output of the LAST launch for one
 * (stream, chain) to host: out[2*k], out[2*k+1] = I,Q; returns sample count
 * (for a stream ended by nvx_finish: the samples its real input produced)     */
NVX_API size_t nvx_debug_y3(nvx_handle *h, int stream, int chain, double *out, size_t cap_pairs);
/* debug tap: discriminator output (delta-phi) of the last launch             */
NVX_API size_t nvx_debug_dphi(nvx_handle *h, int stream, int chain, double *out, size_t cap);

/* device memory helpers so callers need no HIP bindings of their own         */
NVX_API int   nvx_device_count(void);
NVX_API void *nvx_device_alloc(int device, size_t bytes);
NVX_API void  nvx_device_free(int device, void *p);
NVX_API int   nvx_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes);
NVX_API int   nvx_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes);
NVX_API int   nvx_device_sync(int device);
/* a hipStream_t of the caller's own (for nvx_process_resident's hip_stream argument), and its release          */
NVX_API void *nvx_stream_create(int device);
NVX_API void  nvx_stream_destroy(int device, void *hip_stream);

/* ==========================================================================
 * C'. Several GPUs behind one object.  NAVTEX chains share no state (receiver/decoder.h:31-60,
 *    receiver/nav_b_sm.h:92-114; FIR1 + mixer state per stream: receiver/fir1cpp.C:57-60, receiver/fir2cpp.C:74-83),
 *    so streams shard one contiguous subset per device and nothing is exchanged between devices: no collective.
 *    A group owns one handle and one host thread per member; global stream id g lives on member m with
 *    first(m) <= g < first(m) + count(m), first(m) = m * (S / n) + min(m, S % n).  The two chains of a stream stay
 *    together (they share FIR1).  cfg is read as for nvx_create with n_streams = the TOTAL S; chain_masks / labels
 *    are indexed by global stream; cfg.device is ignored; on_message receives the GLOBAL stream id (wideband mode:
 *    members own INPUT streams, decoded streams keep their meaning 8 * w + k with w the global input stream; that
 *    is the id in messages, nvx_group_poll_bits and nvx_group_bit_count).  Messages are
 *    delivered by the thread that calls nvx_group_fetch_bits / nvx_group_flush, member after member, stream after
 *    stream -- the same order one handle of S streams would use.  Two members may name the same device.
 * ========================================================================== */
typedef struct nvx_group nvx_group;
NVX_API int    nvx_group_create(const int *devices, int n_members, const nvx_config *cfg, nvx_group **out);
NVX_API void   nvx_group_destroy(nvx_group *g);
NVX_API int    nvx_group_reset(nvx_group *g);
NVX_API int    nvx_group_size(const nvx_group *g);
/* member m: its device, first global stream, stream count and handle (any of the out pointers may be NULL)   */
NVX_API int    nvx_group_member(nvx_group *g, int m, int *device, int *first_stream, int *n_streams, nvx_handle **h);
/* global stream id -> member index (or -1)                                                                   */
NVX_API int    nvx_group_member_of(const nvx_group *g, int global_stream);
/* device-resident input: d_iq[m] is member m's buffer ON ITS DEVICE, layout [count(m)][pitch] as for
 * nvx_process_resident.  The launches are issued by the members' own threads, all devices at once; returns
 * after they have been QUEUED.  Errors surface at the next fetch.                                             */
NVX_API int    nvx_group_process_resident(nvx_group *g, const void *const *d_iq, size_t pitch_samples,
                                          size_t first_frame, int n_frames);
/* wait for every member, run the character layers (in parallel, one member per thread), deliver messages      */
NVX_API int    nvx_group_fetch_bits(nvx_group *g);
/* host input by global stream id (nvx_push_iq of the owning member; cfg.push_mode) and the matching flush     */
NVX_API int    nvx_group_push_iq(nvx_group *g, int global_stream, const int16_t *iq_interleaved, size_t n);
NVX_API int    nvx_group_flush(nvx_group *g);
/* nvx_finish of every member (end of the input of every stream), then the messages as nvx_group_flush delivers them */
NVX_API int    nvx_group_finish(nvx_group *g);
NVX_API size_t nvx_group_poll_bits(nvx_group *g, int global_stream, int chain, char *out, size_t cap);
NVX_API size_t nvx_group_bit_count(nvx_group *g, int global_stream, int chain);
/* Bind the CALLING thread to the CPUs of the NUMA node the HIP device hangs off (PCI bus id -> sysfs numa_node /
 * local_cpulist).  Returns the number of CPUs bound to, 0 when the platform gives no answer (affinity unchanged),
 * or a negative error.  The group calls it for its member threads unless NVX_GROUP_NUMA=0.                     */
NVX_API int    nvx_bind_thread_to_device(int device);

/* ==========================================================================
 * D. Host SITOR-B / CCIR-476 character layer
 *    (receiver/nav_b_sm.h:56-128, receiver/nav_b_sm.C).  Stand-alone C; used
 *    internally by C and exported for callers that bring their own bits.
 * ========================================================================== */
typedef struct nvx_sitor nvx_sitor;
typedef void (*nvx_sitor_msg_fn)(void *user, const char *bbbb, const char *message, int freq);
/* trace sink: receives the text the reference prints to stdout (may be NULL) */
typedef void (*nvx_sitor_trace_fn)(void *user, const char *text);
NVX_API nvx_sitor *nvx_sitor_new(int freq, nvx_sitor_msg_fn on_msg, void *user);
NVX_API void nvx_sitor_set_trace(nvx_sitor *s, nvx_sitor_trace_fn fn, void *user);
NVX_API void nvx_sitor_free(nvx_sitor *s);
NVX_API void nvx_sitor_reset(nvx_sitor *s);
NVX_API void nvx_sitor_receive_bit(nvx_sitor *s, char bit);            /* nav_b_sm.C:266 */
NVX_API void nvx_sitor_receive_bits(nvx_sitor *s, const char *bits, size_t n);

/* ==========================================================================
 * E. WAV file path (receiver/wav.h:129-217 subset, 44-byte canonical PCM
 *    header as receiver/wav.c writes it; 2 channels = I,Q, 16 bit, 252 kHz,
 *    receiver/capt_sched.c:87-96).
 * ========================================================================== */
typedef struct nvx_wav nvx_wav;
#define NVX_WAV_OPEN_READ  1
#define NVX_WAV_OPEN_WRITE 2
NVX_API nvx_wav *nvx_wav_open(const char *filename, uint32_t mode);    /* wav.h:129 */
NVX_API int      nvx_wav_close(nvx_wav *w);                             /* wav.h:130 */
NVX_API size_t   nvx_wav_read(nvx_wav *w, void *buffer, size_t frames); /* wav.h:141 */
NVX_API size_t   nvx_wav_write(nvx_wav *w, const void *buffer, size_t frames); /* wav.h:151 */
NVX_API void     nvx_wav_set_format(nvx_wav *w, uint16_t format);       /* wav.h:178 */
NVX_API void     nvx_wav_set_num_channels(nvx_wav *w, uint16_t n);      /* wav.h:186 */
NVX_API void     nvx_wav_set_sample_rate(nvx_wav *w, uint32_t rate);    /* wav.h:194 */
NVX_API void     nvx_wav_set_sample_size(nvx_wav *w, size_t bytes);     /* wav.h:210 */
NVX_API uint16_t nvx_wav_get_format(const nvx_wav *w);                  /* wav.h:212 */
NVX_API uint16_t nvx_wav_get_num_channels(const nvx_wav *w);            /* wav.h:213 */
NVX_API uint32_t nvx_wav_get_sample_rate(const nvx_wav *w);             /* wav.h:214 */
NVX_API size_t   nvx_wav_get_sample_size(const nvx_wav *w);             /* wav.h:216 */
NVX_API size_t   nvx_wav_get_length(const nvx_wav *w);                  /* wav.h:217, frames */
NVX_API const char *nvx_wav_err(void);                                  /* wav.h:114 */
/* File harness the reference lacks: open -> loop wav_read ->
 * the capt_sched.c:509-513 loop, on the GPU, ended by nvx_stream_finish: the
 * bits are the reference's on the same file, whatever its length; the stream is
 * ended afterwards (only an empty file on a fresh stream ends nothing) and
 * nvx_stream_reset / nvx_reset start a new one.  Returns the number of frames
 * (the last one may be partial) or a negative error.                           */
NVX_API int nvx_decode_wav(nvx_handle *h, int stream, const char *filename);

/* ==========================================================================
 * F. Deterministic synthetic source (no reference counterpart).
 *    Integer-only CPFSK modulator, bit-identical on host and device.
 * ========================================================================== */
/* SITOR-B transmit framing of `text` (A-Z 0-9 space and the CCIR-476 figure
 * set, '\n' = CR LF): n_phasing (DX=0x4C, RX=0x07) pairs, DX/RX time diversity
 * (RX repeats the DX character of two pairs earlier), three closing idle
 * pairs.  Writes one byte per bit, 'B' or 'Y'; returns the bit count, or the
 * needed capacity when bits == NULL.                                         */
NVX_API size_t nvx_sitor_encode(const char *text, int n_phasing, char *bits, size_t cap);

typedef struct nvx_carrier {
    int32_t  freq_hz;        /* carrier offset from the stream centre (+14000 / -14000)   */
    int32_t  shift_hz;       /* 'B' = freq+shift, 'Y' = freq-shift (85)                   */
    int32_t  amplitude;      /* LSB                                                       */
    uint32_t phase0;         /* initial phase, 2^32 = one turn                            */
    uint32_t bit_offset;     /* samples by which bit boundaries lead n = 0, < samples/bit */
    uint32_t n_bits;         /* bits[] length; the sequence repeats                       */
    const char *bits;        /* 'B'/'Y' (host pointer)                                    */
} nvx_carrier;

#define NVX_SYNTH_MAX_CARRIERS 16     /* a wideband stream: 8 sub-bands x 2 chains             */
typedef struct nvx_synth_stream {
    uint32_t seed;
    int32_t  noise_amp;      /* uniform integer noise in [-noise_amp, +noise_amp]         */
    int32_t  n_carriers;     /* 0..NVX_SYNTH_MAX_CARRIERS                                 */
    nvx_carrier carrier[NVX_SYNTH_MAX_CARRIERS];
} nvx_synth_stream;

/* host generator: n complex samples starting at sample index n0, sample_rate
 * NVX_RATE_RAW or NVX_RATE_IN, into out[2*n]                                 */
NVX_API int nvx_synth_host(const nvx_synth_stream *s, uint32_t sample_rate,
                           uint64_t n0, size_t n, int16_t *out_iq);
/* device generator: fills d_out [n_streams][pitch] (complex samples) for
 * sample indices [0, n) of every stream                                      */
NVX_API int nvx_synth_device(int device, const nvx_synth_stream *streams, int n_streams,
                             uint32_t sample_rate, size_t n, void *d_out, size_t pitch_samples);

/* ==========================================================================
 * G. Wideband front-end (no reference counterpart: the reference tunes ONE 252 kS/s
 *    slice, receiver/capt_sched.c:356-417).  An 8-channel polyphase
 *    channeliser in integer arithmetic: one 2.016 MS/s stream -> eight 252 kS/s
 *    sub-bands centred at k * 252 kHz (k = 4..7 are the negative frequencies), each a
 *    valid input stream of a raw_rate = 0 handle (two NAVTEX chains at +-14 kHz per
 *    sub-band -> up to 16 carriers per wideband stream).
 *    d_raw: [n_wide][pitch_raw] packed IQ; n_out outputs per sub-band (multiple of 64)
 *    are produced from the 8*n_out raw samples starting at first_sample;
 *    d_sub: [n_wide*8][pitch_sub], written from column sub_first.
 *    d_hist_in / d_hist_out: [n_wide][40] packed raw samples carried between calls
 *    (NULL in = silence before the first sample; NULL out = not saved; in != out).
 * ========================================================================== */
#define NVX_WB_SUBBANDS 8
/* the handle's own HIP stream (a hipStream_t), so a channeliser launch can be ordered in front of
 * nvx_process_resident(..., hip_stream = that stream) without a device-wide synchronise      */
NVX_API void *nvx_handle_stream(nvx_handle *h);
/* optional HIP-event timing of the channeliser launches (off by default)                  */
NVX_API void nvx_channelise_timing(int enable);
NVX_API int  nvx_channelise_time_stats(double *sum_ms, uint64_t *launches, int reset);
NVX_API int nvx_channelise_resident(int device, const void *d_raw, size_t pitch_raw, size_t first_sample,
                                    int n_wide, size_t n_out, const void *d_hist_in, void *d_hist_out,
                                    void *d_sub, size_t pitch_sub, size_t sub_first, void *hip_stream);

/* ==========================================================================
 * H. SQLite message sink: writes decoded messages into the
 *    database the reference's web server reads, with the semantics of the
 *    reference's add_message (receiver/message_store.c:59-97: a repeat of the
 *    same bbbb replaces the earlier row; timestamp = UTC "%Y-%m-%d %H:%M";
 *    age = 'NEW') and its schema (receiver/generate_db.sql:3-8).  libsqlite3 is
 *    bound at run time; NVX_ERR_IO if it is not installed.
 *    When the library's own weak add_message is the sink (nothing else linked
 *    defines it) and NAVTEX_AMD_DB names a database file, messages go there.
 * ========================================================================== */
typedef struct nvx_store nvx_store;
/* create_schema != 0: create the messages/config tables if they are missing  */
NVX_API int  nvx_store_open(const char *path, int create_schema, nvx_store **out);
NVX_API void nvx_store_close(nvx_store *s);
/* same arguments and return values as add_message: 0, -1 (open failed), -2 (insert failed) */
NVX_API int  nvx_store_add_message(nvx_store *s, const char *bbbb, const char *message, int freq);
/* an nvx_message_fn: set cfg.on_message = nvx_store_on_message, cfg.user = the store */
NVX_API void nvx_store_on_message(void *user, int stream, const char *bbbb, const char *message, int freq);
/* purge_old_messages (receiver/message_store.c:220-262): delete rows older than
 * max_age_seconds (<= 0: the reference's 72 h); returns the number deleted or < 0 */
NVX_API int  nvx_store_purge(nvx_store *s, long max_age_seconds);
NVX_API void nvx_store_stats(nvx_store *s, uint64_t *added, uint64_t *failed);
/* test hook: a fixed clock (unix seconds) for timestamps and purge; 0 = wall clock */
NVX_API void nvx_store_set_time(nvx_store *s, int64_t unix_seconds);

#ifdef __cplusplus
}
#endif
#endif /* NAVTEX_AMD_H */
