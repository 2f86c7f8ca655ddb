// nvx_fsm_host.cpp -- host side of the demodulator FSM tables (nvx_fsm.h): generation and self-test.
#include "nvx_handle.h"
#include "nvx_fsm.h"

// Bit-period transition table of the demodulator FSM, generated once from the per-sample rule (nvx_fsm.h).
const uint32_t *nvx_fsm_table_host()
{
    static const std::vector<uint32_t> table = [] {
        std::vector<uint32_t> t(NVX_FSM_TABLE_ALLOC, 0u);
        for (int p1 = 0; p1 < 9; p1++)
            for (int so = 0; so < 10; so++)
                for (int a = 0; a < 9; a++)
                    for (int b = 0; b < 9; b++) t[NVX_FSM_KEY(p1, so, a, b)] = nvx_fsm_table_entry(p1, so, a, b);
        for (int prev1 = 0; prev1 < 10; prev1++)
            for (int rawc = 0; rawc < 10; rawc++) t[NVX_FSM_TIMING_BASE + prev1 * 10 + rawc] = nvx_fsm_timing_entry(prev1, rawc);
        return t;
    }();
    return table.data();
}

// Replays pseudo-random front-kernel words through the per-sample rule and through the table, and counts
// differences in the decided bits and in the carried registers.  No device needed (tests/test_host_layer.py).
extern "C" int nvx_fsm_selftest(uint32_t seed, int periods)
{
    const uint32_t *tab = nvx_fsm_table_host();
    uint32_t x = seed ? seed : 1u;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    // per-sample registers (the reference's variables) and per-period registers
    int synced = 0, sync_off = 0, next_sync_off = 0, phase = -1, prev_a = -1;
    nvx_fsm_regs r = { 0, NVX_FSM_UNSYNCED, 0, -1 };
    const int lead = (int)(rnd() % 70u);                     // periods before the class sums are primed
    int raw = (int)(rnd() % 9u), bad = 0;
    for (int m = 0; m < periods; m++) {
        const uint32_t u = rnd();
        if (u % 7u == 0) raw = (int)((u >> 8) % 9u);           // timing jumps; otherwise it drifts or holds
        else if (u % 7u == 1) raw = (raw + 1) % 9;
        else if (u % 7u == 2) raw = (raw + 8) % 9;
        const unsigned w = ((u >> 16) & 0x1ffu) | ((unsigned)(m < lead ? 15 : raw) << 12);
        unsigned want = 0; int n_want = 0;
        for (int k = 0; k < 9; k++) {
            if (k == NVX_FSM_TIMING_SAMPLE) {
                int offset;
                const int have = nvx_fsm_timing((int)(w >> 12), &prev_a, &offset);
                sync_off = (have && !synced) ? offset : sync_off;      // decoder.C:62-70
                next_sync_off = have ? offset : next_sync_off;
                synced = have ? 1 : synced;
            }
            if (nvx_fsm_bit_step(k, synced, &phase, &sync_off, next_sync_off)) { want |= ((w >> k) & 1u) << n_want; n_want++; }
        }
        // the sample-by-sample form the kernel uses in the period in which a stream ENDS, carried through all nine samples:
        // the same bits and registers as the table
        nvx_fsm_regs rp = r;
        int n_part;
        const unsigned part = nvx_fsm_partial_period(w, 9, &rp, &n_part);
        int n_got;
        const unsigned got = nvx_fsm_period(tab, w, &r, &n_got) & ((1u << n_got) - 1u);
        if (n_part != n_got || part != got || rp.phase1 != r.phase1 || rp.so != r.so || rp.nso != r.nso || rp.prev_offset != r.prev_offset) bad++;
        if (n_got != n_want || got != want) bad++;
        if (r.phase1 != phase + 1 || r.nso != next_sync_off || r.prev_offset != prev_a ||
            (r.so != NVX_FSM_UNSYNCED) != (synced != 0) || (synced && r.so != sync_off)) bad++;
    }
    return bad;
}
