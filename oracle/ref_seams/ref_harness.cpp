// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Link-seam harness around the *unmodified* reference sources under
// /root/reference/receiver (compiled where they lie by oracle/Makefile; the
// objects land in oracle/_ref/ which is git-ignored).  Each SEAM_* build swaps
// exactly one downstream reference object for a recorder with the same
// interface, so the double / bit / message stream crossing that seam can be
// captured and committed as a golden vector (tests/golden/make_golden.py).
//
//   SEAM_FULL  fir1+fir2+fir3+decoder+nav_b_sm+nav_sched, add_message captured
//   SEAM_BITS  same, nav_b_sm replaced by a recorder    -> 'B'/'Y' per chain
//   SEAM_FIR1  fir1cpp.o only, sample_in_2 recorded     -> y1 (63 kS/s)
//   SEAM_FIR2  fir1+fir2, fir_filter3 recorded          -> y2 per chain (9 kS/s)
//   SEAM_FIR3  fir1+fir2+fir3, decoder recorded         -> y3 per chain (900 S/s)
//   SEAM_DEC   decoder.o only, fed doubles, bsm recorded-> bits
//   SEAM_SM    nav_b_sm.o only, fed bits                -> messages + stdout
//
// usage: ref_<seam> <input.bin> <output-prefix> [probe ...]
//   input  : interleaved int16 I,Q (FULL/BITS/FIR*), interleaved double I,Q
//            (DEC) or ASCII 'B'/'Y' characters (SM)
//   output : <prefix>.<name>.bin files described next to each writer below.
// The reference's own printf tracing goes to stdout untouched.
//
// Two probes of behaviour the product deliberately does NOT copy ("Deviations"):
//   ref_full | ref_bits  in out reinit <n> <which>
//       call the reference's init functions again in front of input sample n, the way a second
//       init_dsp() / init_fir2_wrapper() of capt_sched.c (:552-555, :612) would:
//       which = 1 init_fir_filter1(), 2 init_fir2_wrapper(), 3 both, in that order.
//   ref_dec              in out inject <n> <value>
//       set the decoder's private `int bd_seq_nbr` (decoder.h:60) to <value> in front of 900 S/s
//       sample n -- what 27.6 days of running bring about by themselves (decoder.C:75: it is
//       incremented for ever).  The member is reached by giving THIS translation unit a public view
//       of the class; decoder.o is the unmodified reference object and the layout is the same.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>

static std::string g_prefix;

static std::vector<char> read_all(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<char> buf;
    char tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return buf;
}

static void write_all(const std::string &name, const void *p, size_t bytes)
{
    std::string path = g_prefix + "." + name + ".bin";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (bytes) fwrite(p, 1, bytes, f);
    fclose(f);
}

// ---------------------------------------------------------------- recorders
#if defined(SEAM_SM)
#include "nav_b_sm.h"
#endif
#if defined(SEAM_FULL) || defined(SEAM_SM)
// add_message(bbbb, message, freq): receiver/message_store.h:7.
// record = int32 freq, int32 len_bbbb, bytes, int32 len_msg, bytes
static std::vector<char> g_msgs;
static void put_i32(int32_t v) { g_msgs.insert(g_msgs.end(), (char *)&v, (char *)&v + 4); }
extern "C" int add_message(char *bbbb, char *message, int freq)
{
    put_i32(freq);
    put_i32((int32_t)strlen(bbbb));
    g_msgs.insert(g_msgs.end(), bbbb, bbbb + strlen(bbbb));
    put_i32((int32_t)strlen(message));
    g_msgs.insert(g_msgs.end(), message, message + strlen(message));
    return 0;
}
#endif

#if defined(SEAM_DEC)
#define private public             /* the inject probe (above): this TU only; decoder.o is the reference's own */
#include "decoder.h"
#undef private
#endif
#if defined(SEAM_BITS) || defined(SEAM_DEC)
#include "decoder.h"
static std::vector<char> g_bits518, g_bits490;
// The recorder keeps the frequency in the class's own `freq` member.
byte_state_machine::byte_state_machine(unsigned int frequency) { freq = frequency; }
void byte_state_machine::receive_bit(char b)
{
    (freq == 490 ? g_bits490 : g_bits518).push_back(b);
}
#endif

#if defined(SEAM_FIR1)
static std::vector<double> g_y1;
void sample_in_2(double i, double q) { g_y1.push_back(i); g_y1.push_back(q); }
#endif

#if defined(SEAM_FIR2)
#include "fir3cpp.h"
static std::vector<double> g_y2[2];
static fir_filter3 *g_first = nullptr;
fir_filter3::fir_filter3(decoder *dec) { output_dec = dec; if (!g_first) g_first = this; }
void fir_filter3::sample_in(double i, double q)
{
    std::vector<double> &v = g_y2[this == g_first ? 0 : 1];
    v.push_back(i); v.push_back(q);
}
#endif

#if defined(SEAM_FIR3)
#include "decoder.h"
static std::vector<double> g_y3[2];
static decoder *g_firstdec = nullptr;
decoder::decoder(byte_state_machine *bsm) { output_bsm = bsm; if (!g_firstdec) g_firstdec = this; }
void decoder::sample_in(double i, double q)
{
    std::vector<double> &v = g_y3[this == g_firstdec ? 0 : 1];
    v.push_back(i); v.push_back(q);
}
// nav_b_sm.o is not linked in this seam; give its class the two symbols
// nav_sched-style construction would need.
byte_state_machine::byte_state_machine(unsigned int frequency) { freq = frequency; }
void byte_state_machine::receive_bit(char) {}
#endif

// ------------------------------------------------------------------ drivers
#if defined(SEAM_FULL) || defined(SEAM_BITS) || defined(SEAM_FIR1) || defined(SEAM_FIR2) || defined(SEAM_FIR3)
extern "C" void init_fir_filter1();
extern "C" void sample_in_1(double, double);
#endif
#if defined(SEAM_FULL) || defined(SEAM_BITS)
extern "C" void init_fir2_wrapper();
#endif
#if defined(SEAM_FIR2) || defined(SEAM_FIR3)
#include "fir2cpp.h"
#endif

int main(int argc, char **argv)
{
    if (argc != 3 && argc != 6) { fprintf(stderr, "usage: %s input.bin out-prefix [reinit n which | inject n value]\n", argv[0]); return 2; }
    g_prefix = argv[2];
    std::vector<char> in = read_all(argv[1]);
    const std::string probe = argc == 6 ? argv[3] : "";
    const size_t probe_at = argc == 6 ? (size_t)strtoull(argv[4], nullptr, 10) : (size_t)-1;
    const long probe_arg = argc == 6 ? strtol(argv[5], nullptr, 10) : 0;
    (void)probe_at; (void)probe_arg;

#if defined(SEAM_FULL) || defined(SEAM_BITS) || defined(SEAM_FIR1) || defined(SEAM_FIR2) || defined(SEAM_FIR3)
    const int16_t *iq = (const int16_t *)in.data();
    size_t npairs = in.size() / 4;
#if defined(SEAM_FIR2)
    static fir_filter3 f518(nullptr), f490(nullptr);      // same order as nav_sched.C:16-17
    init_fir_filter1();
    init_fir_filter2(&f518, &f490);
#elif defined(SEAM_FIR3)
    static byte_state_machine sm518(518), sm490(490);
    static decoder d518(&sm518), d490(&sm490);
    static fir_filter3 f518(&d518), f490(&d490);
    init_fir_filter1();
    init_fir_filter2(&f518, &f490);
#elif defined(SEAM_FIR1)
    init_fir_filter1();
#else
    init_fir_filter1();                                    // capt_sched.c:552-555
    init_fir2_wrapper();                                   // capt_sched.c:612
#endif
    for (size_t n = 0; n < npairs; n++) {                  // capt_sched.c:509-513
#if defined(SEAM_FULL) || defined(SEAM_BITS)
        if (probe == "reinit" && n == probe_at) {
            if (probe_arg & 1) init_fir_filter1();
            if (probe_arg & 2) init_fir2_wrapper();
        }
#endif
        sample_in_1((double)iq[2 * n], (double)iq[2 * n + 1]);
    }
#endif

#if defined(SEAM_DEC)
    static byte_state_machine sm(518);
    static decoder dec(&sm);
    const double *y3 = (const double *)in.data();
    size_t n3 = in.size() / 16;
    for (size_t n = 0; n < n3; n++) {
        if (probe == "inject" && n == probe_at) dec.bd_seq_nbr = (int)probe_arg;
        dec.sample_in(y3[2 * n], y3[2 * n + 1]);
    }
#endif

#if defined(SEAM_SM)
    static byte_state_machine sm(518);
    for (size_t n = 0; n < in.size(); n++)
        if (in[n] == 'B' || in[n] == 'Y') sm.receive_bit(in[n]);
#endif

    fflush(stdout);
#if defined(SEAM_FULL) || defined(SEAM_SM)
    write_all("messages", g_msgs.data(), g_msgs.size());
#endif
#if defined(SEAM_BITS) || defined(SEAM_DEC)
    write_all("bits518", g_bits518.data(), g_bits518.size());
    write_all("bits490", g_bits490.data(), g_bits490.size());
#endif
#if defined(SEAM_FIR1)
    write_all("y1", g_y1.data(), g_y1.size() * 8);
#endif
#if defined(SEAM_FIR2)
    write_all("y2_518", g_y2[0].data(), g_y2[0].size() * 8);
    write_all("y2_490", g_y2[1].data(), g_y2[1].size() * 8);
#endif
#if defined(SEAM_FIR3)
    write_all("y3_518", g_y3[0].data(), g_y3[0].size() * 8);
    write_all("y3_490", g_y3[1].data(), g_y3[1].size() * 8);
#endif
    return 0;
}
