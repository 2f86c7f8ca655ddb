/* nvx_sitor.c -- host-side SITOR-B / CCIR-476 character layer and transmit
 * framing.  Plain C, no GPU, no libm, no regex library.
 *
 * Receive side = the behaviour of the reference's class byte_state_machine
 * (receiver/nav_b_sm.h:56-128, receiver/nav_b_sm.C), kept on the host as the
 * north star prescribes: the GPU hands over 100 bit/s per chain, this layer
 * turns them into add_message(bbbb, text, freq) calls.  Table driven:
 *   - phasing detector: one 30-character pattern + a match counter instead of
 *     the reference's 30-case switch (nav_b_sm.C:301-631);
 *   - start/end-of-message framing: two small hand-written matchers with the
 *     leftmost semantics of the POSIX EREs the reference compiles per line
 *     (nav_b_sm.C:69, :82).
 * Observable behaviour preserved on purpose (tests compare with the compiled
 * reference): mismatch in the phasing detector drops to state 0 without
 * re-examining the bit; six-B state absorbs further B's; the detector is
 * muted for 1100 bits after a hit; RX preferred over DX over '*'; three idle
 * signals in a row in the DX slot end the emission; more than 12 bad codes in
 * the last 20 abort; bbbb keeps the FIRST id seen until the message closes;
 * code 0x5C decodes to ' ' and 0x19 to '-' in both shifts.
 */
#include "navtex_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NVX_LINE_MAX 5000            /* nav_b_sm.h:97-98 buffer sizes          */
#define NVX_ERRWIN   20              /* nav_b_sm.h:49 E_BUFFER_SIZE            */
#define NVX_ERRMAX   12              /* nav_b_sm.h:50 ERROR_THRESHOLD          */
#define NVX_MUTE_BITS 1100           /* nav_b_sm.h:52                          */
#define NVX_ALPHA 0x07               /* idle / phasing signal 1, RX slot (nav_b_sm.h:89) */
#define NVX_BETA  0x4c               /* phasing signal 2, DX slot (nav_b_sm.h:90)        */

/* CCIR-476 code -> character for the two shifts (nav_b_sm.h:60-83).
 * '_' = not a valid code; control pseudo characters: l letter shift,
 * f figure shift, n line feed, r carriage return, p alpha, q beta.          */
static const char nvx_ltrs[128 + 1] =
    "_______p___J_WA____F_YS__-D_Z______C_PI__GR_L____MN_H___O_______"
    "___K_QU__fE_q____Xl_____B___ ____V _n___T_______r_______________";
static const char nvx_figs[128 + 1] =
    "_______p___b_2-____*_6'__-%_+ _____:_08__*4_)____.,_*___9_______"
    "___(_17__f3_q____/l_____?___ ____= _n___5_______r_______________";

static const char nvx_phasing[] = "BBBBBBYYYYBBYYBBBBBBYYYYBBYYBB";   /* nav_b_sm.h:10-39 */

enum { SLOT_SEARCH = 0, SLOT_EXPECT_DX, SLOT_EXPECT_RX };

struct nvx_sitor {
    int freq;
    nvx_sitor_msg_fn on_msg; void *user;
    nvx_sitor_trace_fn on_trace; void *trace_user;
    /* bit level */
    int enabled, nbits, mute, matched;
    unsigned acc;
    /* character level */
    int slot, figures;
    unsigned char dx[3]; int dx_pos, dx_full;
    int idle_run, prev_dx_idle;
    char errwin[NVX_ERRWIN]; int err_pos, err_full, err_count;
    /* message level */
    int in_message;
    char line[NVX_LINE_MAX], text[NVX_LINE_MAX], bbbb[10];
    size_t line_len, text_len;
};

static void trace(nvx_sitor *s, const char *t) { if (s->on_trace) s->on_trace(s->trace_user, t); }

static void append(char *buf, size_t *len, const char *src, size_t n)
{
    /* the reference's fixed buffers overflow on pathological input; truncate */
    if (*len + n > NVX_LINE_MAX - 1) n = NVX_LINE_MAX - 1 - *len;
    memcpy(buf + *len, src, n);
    *len += n;
    buf[*len] = 0;
}

void nvx_sitor_reset(nvx_sitor *s)                         /* nav_b_sm.C:16-42 init() */
{
    if (!s) return;                                        /* (every exported entry point takes a NULL object as a no-op) */
    s->matched = 0; s->slot = SLOT_SEARCH; s->figures = 0;
    s->nbits = 0; s->dx_pos = 0; s->dx_full = 0;
    s->err_count = 0; s->err_pos = 0; s->err_full = 0;
    s->idle_run = 0; s->prev_dx_idle = 0;
    s->line[0] = 0; s->text[0] = 0; s->bbbb[0] = 0; s->line_len = 0; s->text_len = 0;
    s->mute = 0; s->enabled = 0; s->in_message = 0;
    /* acc (temp_byte) is deliberately not cleared here, as in the reference;
     * it is cleared on every phasing hit before it is ever used again.       */
}

nvx_sitor *nvx_sitor_new(int freq, nvx_sitor_msg_fn on_msg, void *user)
{
    nvx_sitor *s = (nvx_sitor *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->freq = freq; s->on_msg = on_msg; s->user = user;
    nvx_sitor_reset(s);
    return s;
}
void nvx_sitor_set_trace(nvx_sitor *s, nvx_sitor_trace_fn fn, void *user) { if (!s) return; s->on_trace = fn; s->trace_user = user; }
void nvx_sitor_free(nvx_sitor *s) { free(s); }

static void abort_message(nvx_sitor *s)                    /* nav_b_sm.C:44-52 */
{
    trace(s, "message abort\n");
    if (s->in_message && s->on_msg) s->on_msg(s->user, s->bbbb, s->text, s->freq);
    nvx_sitor_reset(s);
}

static int is_upper(char c) { return c >= 'A' && c <= 'Z'; }
static int is_digit(char c) { return c >= '0' && c <= '9'; }

/* ERE "(CZC|Z.ZC|ZC.C|ZCZ.) +([A-Z][A-Z])([0-9][0-9])", leftmost match; at a
 * given start at most one alternative can match (CZC needs 'C' first, the
 * others 'Z'), the blank run is maximal because a letter must follow it.    */
static int match_som(const char *l, size_t n, size_t *id_at)
{
    for (size_t p = 0; p < n; p++) {
        size_t q = 0;
        if (p + 3 <= n && l[p] == 'C' && l[p + 1] == 'Z' && l[p + 2] == 'C') q = p + 3;
        else if (p + 4 <= n && l[p] == 'Z' &&
                 ((l[p + 2] == 'Z' && l[p + 3] == 'C') ||             /* Z.ZC */
                  (l[p + 1] == 'C' && l[p + 3] == 'C') ||             /* ZC.C */
                  (l[p + 1] == 'C' && l[p + 2] == 'Z')))              /* ZCZ. */
            q = p + 4;
        if (!q || q >= n || l[q] != ' ') continue;
        while (q < n && l[q] == ' ') q++;
        if (q + 4 <= n && is_upper(l[q]) && is_upper(l[q + 1]) && is_digit(l[q + 2]) && is_digit(l[q + 3])) {
            *id_at = q;
            return 1;
        }
    }
    return 0;
}

/* ERE "NNN.*|N.NN.*|NN.N.*": only existence matters to the caller            */
static int match_eom(const char *l, size_t n)
{
    for (size_t p = 0; p + 3 <= n; p++) {
        if (l[p] != 'N') continue;
        if (l[p + 1] == 'N' && l[p + 2] == 'N') return 1;
        if (p + 4 <= n && l[p + 3] == 'N' && (l[p + 2] == 'N' || l[p + 1] == 'N')) return 1;
    }
    return 0;
}

static void line_feed(nvx_sitor *s)                        /* nav_b_sm.C:56-97 */
{
    size_t at = 0;
    if (s->in_message) {
        append(s->text, &s->text_len, s->line, s->line_len);
        append(s->text, &s->text_len, "\n", 1);
        if (s->on_trace) {
            char buf[NVX_LINE_MAX + 32];
            snprintf(buf, sizeof buf, "line added: %s\n", s->line);
            trace(s, buf);
        }
    }
    if (match_som(s->line, s->line_len, &at)) {
        s->text_len = 0; s->text[0] = 0;
        append(s->text, &s->text_len, s->line, s->line_len);
        append(s->text, &s->text_len, "\n", 1);
        /* the reference strncat()s the new id behind whatever bbbb already
         * holds and cuts at 4: an unterminated earlier message keeps its id  */
        size_t have = strlen(s->bbbb);
        for (size_t i = 0; i < 4 && have < 4; i++) s->bbbb[have++] = s->line[at + i];
        s->bbbb[have] = 0;
        trace(s, "============START OF MESSAGE============ \n");
        s->in_message = 1;
    } else if (match_eom(s->line, s->line_len)) {
        if (s->in_message && s->on_msg) s->on_msg(s->user, s->bbbb, s->text, s->freq);
        s->text_len = 0; s->text[0] = 0; s->bbbb[0] = 0;
        trace(s, "============ END OF MESSAGE ============\n");
        s->in_message = 0;
    }
    s->line_len = 0; s->line[0] = 0;
}

static void emit_code(nvx_sitor *s, unsigned code)         /* nav_b_sm.C:100-145; code 0 = error mark */
{
    if (code == 0) { trace(s, "*"); append(s->line, &s->line_len, "*", 1); return; }
    switch (nvx_ltrs[code]) {
    case 'l': s->figures = 0; break;
    case 'f': s->figures = 1; break;
    case 'n': line_feed(s); break;
    case 'r': case 'p': case 'q': break;
    default: {
        char c = s->figures ? nvx_figs[code] : nvx_ltrs[code];
        trace(s, ".");
        append(s->line, &s->line_len, &c, 1);
        trace(s, ";");
    } }
}

static void receive_code(nvx_sitor *s, unsigned code)      /* nav_b_sm.C:150-262 */
{
    switch (s->slot) {
    case SLOT_SEARCH:                                      /* S_BYTE_WAIT */
        if (code == NVX_ALPHA) s->slot = SLOT_EXPECT_DX;   /* an RX-slot idle: a DX follows */
        if (code == NVX_BETA)  s->slot = SLOT_EXPECT_RX;
        break;
    case SLOT_EXPECT_DX:                                   /* S_BYTE_RECEIVED_RX */
        s->dx[s->dx_pos] = (unsigned char)code;
        if (++s->dx_pos == 3) { s->dx_pos = 0; s->dx_full = 1; }
        if (code == NVX_ALPHA) {
            trace(s, "\n alpha received in DX position\n");
            if (s->prev_dx_idle && ++s->idle_run == 2) {
                trace(s, "\nend of emission detected\n");
                trace(s, "\nstopping reception\n");
                abort_message(s);
                break;                                     /* slot stays SEARCH after the reset */
            }
            s->prev_dx_idle = 1;
        } else {
            s->prev_dx_idle = 0;
        }
        s->slot = SLOT_EXPECT_RX;
        break;
    case SLOT_EXPECT_RX:                                   /* S_BYTE_RECEIVED_DX */
        if (s->dx_full) {
            unsigned twin = s->dx[s->dx_pos];              /* the DX copy sent two pairs earlier */
            if (nvx_ltrs[code] != '_')      emit_code(s, code);
            else if (nvx_ltrs[twin] != '_') emit_code(s, twin);
            else                            emit_code(s, 0);
        }
        s->slot = SLOT_EXPECT_DX;
        break;
    }
    /* sliding window over the validity of the last 20 codes, nav_b_sm.C:235-261 */
    char verdict = nvx_ltrs[code];
    if (s->err_full && s->errwin[s->err_pos] == '_') s->err_count--;
    s->errwin[s->err_pos] = verdict;
    if (verdict == '_') s->err_count++;
    if (++s->err_pos == NVX_ERRWIN) { s->err_pos = 0; s->err_full = 1; }
    if (s->err_count > NVX_ERRMAX) {
        emit_code(s, 0);
        trace(s, "\n error th exceeded \n");
        abort_message(s);
    }
}

void nvx_sitor_receive_bit(nvx_sitor *s, char bit)         /* nav_b_sm.C:266-634 */
{
    if (!s) return;
    if (s->enabled) {
        /* the reference shifts a signed char; only 7 bits are ever collected
         * between two clears, so an unsigned accumulator is equivalent        */
        s->acc = ((s->acc << 1) | (bit == 'Y')) & 0xff;
        if (++s->nbits == 7) {
            receive_code(s, s->acc & 0x7f);
            s->nbits = 0;
            s->acc = 0;
        }
    }
    if (s->mute) {
        if (--s->mute == 0) trace(s, "phase det disable timer expired\n");
        return;
    }
    if (s->matched == 29) {
        if (bit == 'B') {
            s->enabled = 1; s->nbits = 0; s->acc = 0;
            trace(s, "phasing detected\n");
            s->mute = NVX_MUTE_BITS;
        }
        s->matched = 0;
    } else if (bit == nvx_phasing[s->matched]) {
        s->matched++;
    } else if (s->matched != 6) {
        s->matched = 0;
    }
}

void nvx_sitor_receive_bits(nvx_sitor *s, const char *bits, size_t n)
{
    if (!s || !bits) return;
    for (size_t i = 0; i < n; i++)
        if (bits[i] == 'B' || bits[i] == 'Y') nvx_sitor_receive_bit(s, bits[i]);
}

/* ========================================================================== */
/* transmit framing for the synthetic source                                  */
/* ========================================================================== */
/* character -> code, derived from the decode tables above (legal 3-of-7 codes
 * only, so the 0x5C blank quirk is never transmitted)                        */
static int popcount7(unsigned v) { int c = 0; for (int i = 0; i < 7; i++) c += (v >> i) & 1; return c; }

static int find_code(const char *table, char ch)
{
    for (unsigned c = 0; c < 128; c++)
        if (table[c] == ch && popcount7(c) == 3) return (int)c;
    return -1;
}

static size_t put_code(char *bits, size_t cap, size_t at, unsigned code)
{
    for (int i = 6; i >= 0; i--) {                         /* MSB first, 'Y' = 1 (nav_b_sm.C:271-275) */
        if (bits && at < cap) bits[at] = ((code >> i) & 1) ? 'Y' : 'B';
        at++;
    }
    return at;
}

size_t nvx_sitor_encode(const char *text, int n_phasing, char *bits, size_t cap)
{
    /* 1. text -> code sequence with shift characters                         */
    if (!text) return 0;
    size_t n = strlen(text), ncodes = 0;
    unsigned char *codes = (unsigned char *)malloc(3 * n + 16);
    if (!codes) return 0;
    const int LTRS_SHIFT = find_code(nvx_ltrs, 'l'), FIGS_SHIFT = find_code(nvx_ltrs, 'f');
    const int CR = find_code(nvx_ltrs, 'r'), LF = find_code(nvx_ltrs, 'n');
    int figures = 0;
    for (size_t i = 0; i < n; i++) {
        char ch = text[i];
        if (ch == '\n') { codes[ncodes++] = (unsigned char)CR; codes[ncodes++] = (unsigned char)LF; continue; }
        if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A');
        int cl = -1, cf = -1;
        if (ch == ' ') cl = cf = find_code(nvx_ltrs, ' ');
        else {
            if (is_upper(ch)) cl = find_code(nvx_ltrs, ch);
            else if (ch != '_' && ch != '*' && !(ch >= 'a' && ch <= 'z')) cf = find_code(nvx_figs, ch);
            if (ch == '-') { cl = -1; cf = find_code(nvx_figs, '-'); }
        }
        if (cl < 0 && cf < 0) continue;                    /* not transmittable: skipped */
        if (cl >= 0 && cf >= 0) { codes[ncodes++] = (unsigned char)cl; continue; }   /* blank: either shift */
        if (cl >= 0) { if (figures) { codes[ncodes++] = (unsigned char)LTRS_SHIFT; figures = 0; } codes[ncodes++] = (unsigned char)cl; }
        else         { if (!figures) { codes[ncodes++] = (unsigned char)FIGS_SHIFT; figures = 1; } codes[ncodes++] = (unsigned char)cf; }
    }
    /* 2. pairs (DX, RX): phasing, then DX = code j, RX = code j-2 (idle for
     *    j < 2); two flushing pairs carry the last two RX copies with an idle
     *    in DX, a third idle pair completes the end-of-emission signature.   */
    size_t at = 0;
    for (int i = 0; i < n_phasing; i++) { at = put_code(bits, cap, at, NVX_BETA); at = put_code(bits, cap, at, NVX_ALPHA); }
    for (size_t j = 0; j < ncodes + 3; j++) {
        unsigned dx = (j < ncodes) ? codes[j] : NVX_ALPHA;
        unsigned rx = (j >= 2 && j - 2 < ncodes) ? codes[j - 2] : NVX_ALPHA;
        at = put_code(bits, cap, at, dx);
        at = put_code(bits, cap, at, rx);
    }
    free(codes);
    return at;
}
