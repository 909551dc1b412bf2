// lp_kernels_prog.hip -- the scans of progressive (SOF2) JPEGs on the device: ONE WAVE PER SCAN.
//
// Replaces libjpeg-turbo's jdphuff.c decode_mcu_DC_first / decode_mcu_AC_first / decode_mcu_DC_refine / decode_mcu_AC_refine behind
// opencv_decoder_read_data (/root/reference/opencv.cpp:166-171) for well-formed streams; anything irregular is flagged and the image is
// decoded by the host route (lp_prog_host.cpp over lp_jbits.h, which does what libjpeg does with damaged data).
//
// Why a wave and not a lane. A scan is a serial chain (a refinement scan parses differently depending on which coefficients of the block
// at hand are already non-zero, so nothing behind an unknown block index can be parsed), and a batch only offers (images x independent
// scans) chains: a few hundred. The first device version (k_prog_scan, lp_kernels_decode.hip: lane = scan, generic code) ran such a chain
// ~30x slower than a host core. Here the 64 lanes of a wave work for ONE chain:
//   * lanes = 64 BIT OFFSETS. At a refill every lane decodes "the symbol that would start at bit pb + lane" -- one LDS lookup in a
//     12-bit table built at kernel start, extra bits / sign / EOB run included -- into one packed 32-bit entry. The chain itself is
//     scalar: v_readlane the entry at the current offset, a handful of SALU instructions, next offset. No memory access, no LDS access
//     and no per-lane work on the symbol chain; one refill per ~60 consumed bits. The stream words travel in two VGPRs (64 + 64 words,
//     overlapping, the second loaded 1024 bits ahead), so a refill is four v_readlane + a funnel shift away from its bits.
//   * lanes = 64 COEFFICIENTS of the block at hand (zigzag index = lane; the arena stores blocks in zigzag order). An AC refinement
//     needs "the (r + 1)-th still-zero coefficient from k" and "how many non-zero ones lie before it": with the ranks of the zero /
//     non-zero lanes taken once per block (v_mbcnt of the two ballots) the first is ONE v_cmp + s_ff1, the second ONE v_readlane.
//     The correction bits are not read on the chain at all: every symbol only notes where its correction bits start (two VALU
//     instructions: lanes at or behind k take the new base), and at the end of the block every non-zero lane fetches ITS bit from the
//     stream registers (ds_bpermute) and corrects its coefficient -- the device form of the host decoder's PDEP walk.
//   * blocks of an AC refinement are prefetched four ahead (2-byte loads, lane = coefficient), stores touch only the band's lanes: the
//     DC refinement of the same component runs in the same dependency level and owns element 0.
// A lone wave issues one instruction every ~5 cycles whatever the kind, so the cost of a scan is its instruction count: ~25 per AC
// refinement symbol + ~50 per block + ~50 per refill, against the host decoder's ~28 ns per symbol on one core.
//
// Irregular = the image goes to the host route (LpJpegState::error |= LP_PROG_IRREGULAR on the scan's pseudo stream): a code that matches
// nothing, a coefficient index past the band, an AC refinement that runs out of zero coefficients, a read past the end of a restart
// interval (libjpeg's insufficient_data), fewer restart markers than intervals. The unstuff kernels flag wrong marker numbers / counts.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lp_launch.h"
#include "lp_types.h"

namespace {

typedef unsigned long long u64;

#define PW_RARE(x) __builtin_expect(!!(x), 0) // keeps the rare paths out of the straight-line code: a taken branch costs a lone wave ~25 cycles
#define PW_AC_BITS 12u // first-level width of an AC scan's table (8 KB of LDS); longer codes: canonical search, per lane, rare
#define PW_DC_BITS 10u // DC scans: up to four tables (one per scan component) share the same 8 KB

// packed entry of one bit offset
#define PW_ADV(e) ((e) & 63u)           // bits the symbol consumes: code + extra bits (value, sign or EOB run bits)
#define PW_R(e) (((e) >> 6) & 15u)      // run
#define PW_F_ZRL 0x400u                 // AC first scans only (a refinement treats ZRL as a coefficient-less run of 15: the common path)
#define PW_F_EOB 0x800u                 // EOBn
#define PW_F_BAD 0x1000u                // no code matches
#define PW_SPECIAL(e) ((e) & 0x1c00u)   // anything but a coefficient: off the symbol loop's straight path
#define PW_PAY_S(e) ((int32_t)(e) >> 16) // value << Al (first scans), +-(1 << Al) (refinement), DC difference
#define PW_PAY_U(e) ((e) >> 16)          // EOB run


__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// lane `lane` of `old` takes v (both wave-uniform). This clang has no writelane builtin; the intrinsic is reached by its IR name and the
// compiler places the lane select in M0 itself.
extern "C" __device__ int lp_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t wl(uint32_t v, uint32_t lane, uint32_t old) { return (uint32_t)lp_llvm_writelane((int)v, (int)lane, (int)old); }
__device__ __forceinline__ uint32_t mbcnt64(u64 m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ u64 ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// Loads whose latency must not be waited for where they are issued: the AC refinement's blocks (four iterations ahead) and the stream
// words (1024 bits ahead). The compiler's own s_waitcnt insertion gives up on this kernel's control flow -- vmcnt(0) at every use, so each
// block waited ~2 us for the prefetch issued one block earlier -- and hand-issued loads into C++ variables do not survive it either: it
// copies an asm's output register (tied or not) into the variable's register right behind the load, before the data is there. So the
// in-flight values live in PHYSICAL registers the compiler never sees as values (v100 ... v104: far above the ~40 it allocates; named
// only in these asm strings and their clobber lists, which also make the kernel's register count cover them), and enter the
// program through "s_waitcnt vmcnt(N); v_mov" pairs. N = hand-issued loads issued since the one being taken: vector memory operations
// return in order on gfx9, and whatever else is issued in between (stores, rare slow paths) only makes the wait stricter.
// scripts/r06_check_prog_isa.py checks the listing: v100-v104 appear in these instructions only.
#define PW_RING_LOAD_I16(R, ptr) asm volatile("global_load_sshort " R ", %0, off" : : "v"(ptr) : "memory", R)
#define PW_RING_LOAD_U32(R, ptr) asm volatile("global_load_dword " R ", %0, off" : : "v"(ptr) : "memory", R)
#define PW_RING_TAKE(n, R, dst) asm volatile("s_waitcnt vmcnt(" #n ")\n\tv_mov_b32 %0, " R : "=v"(dst) : : "memory", R)

// The unstuffed stream of one scan (k_unstuff_*: big-endian words, bit 31 of word 0 first, restart markers cut out and listed).
struct PwBits {
    const uint32_t* words;
    uint32_t cap;      // words that may be read
    uint32_t lane;
    uint32_t wb;       // uniform: word index lane 0 of `cur` holds
    uint32_t cur;      // per lane: word wb + lane; words wb + 32 + lane are in flight in v104 (see PW_RING_*)
    uint32_t pb;       // uniform: bit position of window offset 0

    // (past the capacity: the last word again; a regular stream never consumes such bits, and one that does is flagged by its
    // interval's end check)
    __device__ __forceinline__ const uint32_t* addr(uint32_t i) const { return words + (i < cap ? i : cap - 1u); }
    __device__ __forceinline__ uint32_t ldw(uint32_t i) const { return *addr(i); }
    __device__ __forceinline__ void seek(uint32_t p)
    {
        wb = p >> 5;
        // (the load carries its own wait: a compiler-visible one leaves "cur may be pending" at every join behind it, i.e. a vmcnt(0) --
        // which drains the block prefetches -- at every refill)
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(cur) : "v"(addr(wb + lane)) : "memory");
        PW_RING_LOAD_U32("v104", addr(wb + 32u + lane));
    }
    // 32 words on: the second register (in flight in v104 since the last shift / seek) becomes the first. old_enough: at least eight
    // vector memory operations were issued since (the AC refinement counts its blocks: a load and a store each), so the wait can leave
    // the block prefetches in flight
    __device__ __forceinline__ void shift(bool old_enough)
    {
        if (old_enough) PW_RING_TAKE(7, "v104", cur);
        else PW_RING_TAKE(0, "v104", cur);
        wb += 32u;
        PW_RING_LOAD_U32("v104", addr(wb + 32u + lane));
    }
    // the 32 bits at p + lane, for every lane; sets pb = p
    __device__ __forceinline__ uint32_t window(uint32_t p)
    {
        uint32_t q = (p >> 5) - wb;
        if (q > 60u) { // uniform; the callers shift at block starts (q >= 24), so this is a jump (restart interval) or an absurd block
            if (q < 93u) { shift(false); q -= 32u; }
            else { seek(p); q = 0; }
        }
        pb = p;
        const uint32_t w0 = rl(cur, q), w1 = rl(cur, q + 1u), w2 = rl(cur, q + 2u), w3 = rl(cur, q + 3u);
        const uint32_t t = (p & 31u) + lane, idx = t >> 5;
        const uint32_t hi = idx == 0u ? w0 : idx == 1u ? w1 : w2;
        const uint32_t lo = idx == 0u ? w1 : idx == 1u ? w2 : w3;
        return (uint32_t)(((((u64)hi) << 32) | lo) << (t & 31u) >> 32);
    }
    // bit `bp` of the stream, per lane: from `cur` (bp lies at or behind the start of the block at hand, and the register is only
    // moved on between blocks), else -- a block longer than ~1000 bits -- from memory
    __device__ __forceinline__ uint32_t bit_at(uint32_t bp, bool active)
    {
        const uint32_t wi = (bp >> 5) - wb;
        uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((wi & 63u) << 2), (int)cur);
        const bool far = active && wi >= 64u;
        if (PW_RARE(ballot(far))) { // (the load carries its own wait: a compiler-visible one puts a vmcnt(0) at the join, on every block's path)
            if (far) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(w) : "v"(addr(bp >> 5)) : "memory");
        }
        return (w >> (31u - (bp & 31u))) & 1u;
    }
};

// The canonical part of one table (T.81 F.2.2.3) in LDS, for the codes longer than the lookup width: no vector memory on the slow path
// (its waits would drain the block prefetches).
struct PwCanon {
    int32_t maxcode[18];
    int32_t valoff[18];
    uint8_t vals[256];
};

// lut[i] = (code length << 8) | symbol for the `bits`-bit window i, 0 = a longer code (or none)
__device__ __forceinline__ void pw_build_lut(uint16_t* lut, PwCanon* can, uint32_t bits, const LpProgHuff* ht, uint32_t slot, uint32_t lane)
{
    if (lane < 18u) can->maxcode[lane] = ht->maxcode[slot][lane];
    if (lane < 17u) can->valoff[lane] = ht->valoff[slot][lane];
    for (uint32_t i = lane; i < 256u; i += 64u) can->vals[i] = ht->vals[slot][i];
    for (uint32_t i = lane; i < (1u << bits); i += 64u) {
        uint32_t e = 0;
        for (uint32_t l = 1; l <= bits; l++) {
            const int32_t code = (int32_t)(i >> (bits - l));
            if (code <= ht->maxcode[slot][l]) {
                e = (l << 8) | ht->vals[slot][(uint32_t)(code + ht->valoff[slot][l]) & 255u];
                break;
            }
        }
        lut[i] = (uint16_t)e;
    }
}

// (length << 8) | symbol of the code at the top of `peek`; length 17 = nothing matches (jdhuff.c jpeg_huff_decode's sentinel)
__device__ __forceinline__ uint32_t pw_lookup(const uint16_t* lut, const PwCanon* can, uint32_t bits, uint32_t peek)
{
    uint32_t e = lut[peek >> (32u - bits)];
    if (PW_RARE(ballot(e == 0u))) {
        if (e == 0u) {
            e = 17u << 8;
            for (uint32_t l = bits + 1u; l <= 16u; l++) {
                const int32_t code = (int32_t)(peek >> (32u - l));
                if (code <= can->maxcode[l]) {
                    e = (l << 8) | can->vals[(uint32_t)(code + can->valoff[l]) & 255u];
                    break;
                }
            }
        }
    }
    return e;
}

__device__ __forceinline__ int32_t pw_extend(uint32_t v, uint32_t s) // HUFF_EXTEND, s >= 1
{
    return v < (1u << (s - 1u)) ? (int32_t)v - (int32_t)((1u << s) - 1u) : (int32_t)v;
}

// the packed entry of an AC symbol (first scan or refinement) whose code starts at the top of `peek`
__device__ __forceinline__ uint32_t pw_entry_ac(uint32_t e16, uint32_t peek, bool refine, uint32_t Al)
{
    const uint32_t len = e16 >> 8, sym = e16 & 255u, r = sym >> 4, s = sym & 15u;
    if (len > 16u) return PW_F_BAD;
    const uint32_t rest = peek << len; // what follows the code (len <= 16: at least 16 bits of it)
    if (s) {
        uint32_t adv, pay;
        if (refine) { // a new coefficient: its size is 1 whatever the symbol says (jdphuff.c warns and goes on), then its sign
            adv = len + 1u;
            pay = (rest >> 31) ? (1u << Al) : (uint32_t)(-(int32_t)(1u << Al));
        } else {
            adv = len + s;
            pay = (uint32_t)pw_extend(rest >> (32u - s), s) << Al;
        }
        return adv | (r << 6) | (pay << 16);
    }
    if (r == 15u) return len | (15u << 6) | (refine ? 0u : PW_F_ZRL); // ZRL
    const uint32_t run = (1u << r) + (r ? rest >> (32u - r) : 0u);     // EOBn, n = r <= 14
    return (len + r) | (r << 6) | PW_F_EOB | (run << 16);
}

__device__ __forceinline__ uint32_t pw_entry_dc(uint32_t e16, uint32_t peek)
{
    const uint32_t len = e16 >> 8, t = e16 & 15u;
    if (len > 16u) return PW_F_BAD;
    const uint32_t rest = peek << len;
    const uint32_t diff = t ? (uint32_t)pw_extend(rest >> (32u - t), t) : 0u;
    return (len + t) | (diff << 16);
}

struct PwScan { // what every scan type needs, wave-uniform
    const LpProgScan* sc;
    const LpProgHuff* ht;
    int16_t* coef;      // the image's first block
    const uint32_t* rst;
    uint32_t n_rst, total_bits;
    uint32_t lane;
    uint32_t* err;
    // pipelined launches (every dependency level of the decode range in one grid): this scan's progress word, its dependencies
    uint32_t* progress; // 0 = level-by-level launches: nothing to wait for, nothing to publish
    const LpProgDep* dep;
    uint32_t self;
    __device__ __forceinline__ void irregular() const
    {
        if (lane == 0u) atomicOr(err, LP_PROG_IRREGULAR);
    }
    // MCU rows [0, rows) of this scan are final: visible to the device before the counter moves
    __device__ __forceinline__ void publish(uint32_t rows) const
    {
        if (!progress) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0u) __hip_atomic_store(progress + self, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // before MCU row `my` of this scan is read or written: every scan it refines has finished the block rows it covers. A producer
    // always holds a lower ticket (it is running or done), so the wait ends; the clock is there for the day something else breaks:
    // ~2 s without progress flags the image (host route) and goes on.
    __device__ __forceinline__ bool wait_for(uint32_t my) const
    {
        if (!progress) return true;
        bool ok = true;
        const uint32_t nd = dep->ndep;
        for (uint32_t d = 0; d < nd; d++) {
            const uint32_t need = ((my + 1u) * dep->vs_self[d] - 1u) / dep->vs_dep[d] + 1u; // (a producer that has finished publishes ~0)
            const uint32_t* p = progress + dep->scan[d];
            if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) continue;
            const uint64_t t0 = wall_clock64();
            while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(32);
                if (wall_clock64() - t0 > 200000000ull) { ok = false; break; } // 100 MHz
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return ok;
    }
};

// Restart bookkeeping shared by the four scan types: the interval's end and the jump to the next one.
struct PwIntervals {
    uint32_t dri, left, k, end; // MCUs per interval (0: none), MCUs left in this one, intervals passed, last bit of the current one
    __device__ __forceinline__ void init(const PwScan& s)
    {
        dri = s.sc->dri;
        left = dri;
        k = 0;
        end = (dri && s.n_rst) ? s.rst[0] : s.total_bits;
    }
    // before an MCU: true when a new interval starts here (then *p = its first bit). *bad: no marker left, or the old interval was
    // overrun -- the image goes to the host route; decoding goes on where it stands (harmlessly: every index below is bounded).
    __device__ __forceinline__ bool crossing(const PwScan& s, uint32_t p_now, uint32_t* p, bool* bad)
    {
        if (!dri) return false;
        if (left) { left--; return false; }
        left = dri - 1u;
        const bool ok = p_now <= end && k < s.n_rst;
        *bad = *bad || !ok;
        *p = ok ? s.rst[k < s.n_rst ? k : 0u] : p_now;
        k++;
        end = k < s.n_rst ? s.rst[k] : s.total_bits;
        return true;
    }
};

// ---- decode_mcu_AC_first: one component, band Ss..Se, values << Al, EOB runs
__device__ __forceinline__ void pw_ac_first(const PwScan& s, PwBits& b, const uint16_t* lut, const PwCanon* can)
{
    const LpProgScan& sc = *s.sc;
    const uint32_t Ss = sc.Ss, Se = sc.Se, Al = sc.Al, lane = s.lane;
    PwIntervals iv;
    iv.init(s);
    uint32_t eobrun = 0, o = 0;
    bool bad = false;
    auto refill = [&](uint32_t p) __attribute__((always_inline)) {
        const uint32_t peek = b.window(p);
        o = 0;
        return pw_entry_ac(pw_lookup(lut, can, PW_AC_BITS, peek), peek, false, Al);
    };
    b.seek(0);
    uint32_t E = refill(0);
    for (uint32_t my = 0; my < sc.mcuy; my++) {
        int16_t* row = s.coef + ((size_t)sc.cblk[0] + (size_t)my * sc.bw[0]) * 64u;
        bad = !s.wait_for(my) || bad;
        if (my) s.publish(my);
        for (uint32_t mx = 0; mx < sc.mcux; mx++) {
            uint32_t np;
            if (iv.crossing(s, b.pb + o, &np, &bad)) {
                eobrun = 0;
                E = refill(np);
            }
            if (eobrun) { eobrun--; continue; }
            if (((b.pb + o) >> 5) - b.wb >= 32u && ((b.pb + o) >> 5) - b.wb < 61u) b.shift(false);
            uint32_t c = 0, k = Ss, special = 0;
            while (k <= Se) {
                if (PW_RARE(o >= 64u)) E = refill(b.pb + o);
                const uint32_t e = rl(E, o);
                o += PW_ADV(e);
                if (PW_RARE(PW_SPECIAL(e))) {
                    if (e & PW_F_ZRL) { k += 16u; continue; }
                    special = e;
                    break;
                }
                k += PW_R(e);
                c = wl((uint32_t)PW_PAY_S(e), k & 63u, c);
                k++;
            }
            // an index past the band: jdphuff.c would store outside it (or at natural_order[64..]) -- the host route's case; so is a bad code
            bad = bad || k > Se + 1u || (special & PW_F_BAD);
            if (special & PW_F_EOB) eobrun = PW_PAY_U(special) - 1u;
            if (c != 0u) row[(size_t)mx * 64u + lane] = (int16_t)c; // the arena was zeroed: only what the scan set
        }
    }
    if (bad || b.pb + o > iv.end) s.irregular();
}

// ---- decode_mcu_AC_refine
__device__ __forceinline__ void pw_ac_refine(const PwScan& s, PwBits& b, const uint16_t* lut, const PwCanon* can)
{
    const LpProgScan& sc = *s.sc;
    const uint32_t Ss = sc.Ss, Se = sc.Se, Al = sc.Al, lane = s.lane;
    const int32_t p1 = 1 << Al, m1 = -(1 << Al);
    const u64 band = (Se >= 63u ? ~0ull : (1ull << (Se + 1u)) - 1ull) & (~0ull << Ss);
    const bool in_band = (band >> lane) & 1ull;
    PwIntervals iv;
    iv.init(s);
    uint32_t eobrun = 0, o = 0;
    bool bad = false;
    auto refill = [&](uint32_t p) __attribute__((always_inline)) {
        const uint32_t peek = b.window(p);
        o = 0;
        return pw_entry_ac(pw_lookup(lut, can, PW_AC_BITS, peek), peek, true, Al);
    };
    b.seek(0);
    uint32_t E = refill(0);
    // the blocks in scan order, four loads ahead of the block at hand (past the last block: the last block again -- every iteration
    // issues exactly one load and one store, so that the wait before a block's first use can leave the younger ones in flight)
    const uint32_t nblk = sc.mcux * sc.mcuy, mcux = sc.mcux;
    const size_t row_pitch = (size_t)sc.bw[0] * 64u;
    uint32_t pf_mx = 0, pf_n = 0;
    const int16_t* pf_row = s.coef + (size_t)sc.cblk[0] * 64u;
    uint32_t pf_my = 0, pf_waited = 0, cur_my = 0;
    auto fetch_ptr = [&]() __attribute__((always_inline)) -> const int16_t* {
        if (PW_RARE(s.progress && pf_my >= pf_waited)) { // the cursor enters a row: what it is about to read must be final
            bad = !s.wait_for(pf_my) || bad;
            pf_waited = pf_my + 1u;
        }
        const int16_t* ptr = pf_row + (size_t)pf_mx * 64u + lane;
        if (pf_n + 1u < nblk) {
            pf_n++;
            if (PW_RARE(++pf_mx == mcux)) { pf_mx = 0; pf_row += row_pitch; pf_my++; }
        }
        return ptr;
    };
    uint32_t since_shift = 0; // blocks since the stream registers moved on
    uint32_t cur_mx = 0;
    int16_t* cur_row = s.coef + (size_t)sc.cblk[0] * 64u;
    auto block = [&](int32_t c) __attribute__((always_inline)) { // one block: c = its coefficients (lane = zigzag index)
        int16_t* dst = cur_row + (size_t)cur_mx * 64u + lane;
        if (PW_RARE(cur_mx == 0u && cur_my)) s.publish(cur_my); // the row above is complete (its last store was issued one block ago)
        if (PW_RARE(++cur_mx == mcux)) { cur_mx = 0; cur_row += row_pitch; cur_my++; }
        uint32_t np;
        if (PW_RARE(iv.crossing(s, b.pb + o, &np, &bad))) {
            eobrun = 0;
            E = refill(np);
        }
        // the stream registers move on between blocks only: a block's correction bits are fetched from `cur` at its end
        since_shift++;
        if (PW_RARE(((b.pb + o) >> 5) - b.wb >= 32u && ((b.pb + o) >> 5) - b.wb < 61u)) {
            b.shift(since_shift >= 5u);
            since_shift = 0;
        }
        const u64 nzall = ballot(c != 0);
        const u64 Mb = nzall & band, Zb = ~nzall & band; // the band's coefficients with / without history
        const uint32_t mrank = mbcnt64(Mb);              // non-zero band lanes below me
        const uint32_t zrank = ((Zb >> lane) & 1ull) ? mbcnt64(Zb) : 0xffffu; // my rank among the zero band lanes (only they can match)
        const uint32_t mtotal = (uint32_t)__popcll(Mb);
        // Scalars of the walk: k = next coefficient, zc = zero lanes passed, o = offset of the next symbol in the window, g = o - (non-zero
        // lanes passed). A symbol's correction bits start right behind it, at b.pb + g + its length, for the non-zero lane of rank 0 ...
        uint32_t k = Ss, zc = 0, g = o;
        uint32_t D = 0; // per lane: position of my correction bit - my rank (set by the symbol whose stretch covers me)
        if (eobrun == 0u) {
            uint32_t special = 0, k_special = 0;
            do {
                if (PW_RARE(o >= 64u)) { const uint32_t mc = o - g; E = refill(b.pb + o); g = 0u - mc; }
                const uint32_t e = rl(E, o);
                if (PW_RARE(PW_SPECIAL(e))) { // (no second exit from the loop: two exits cost five scalar instructions per symbol in flag handling)
                    special = e;
                    k_special = k;
                    k = 200u;
                    continue;
                }
                // pass r still-zero coefficients, stop AT the next one: the zero lane of rank zc + r. (None: the stream ran out of zero
                // coefficients and jdphuff.c walks off the band -- the host route's case: stop = 64 ends the block, k = 65 tells)
                const uint32_t t = zc + PW_R(e);
                const u64 hit = ballot(zrank == t);
                const uint32_t stop = hit ? (uint32_t)__builtin_ctzll(hit) : 64u;
                // non-zero lanes below stop: on the scalar side (s_bfm + s_and + s_bcnt1) -- a v_readlane of the lanes' ranks would be one more
                // SALU -> VALU -> SALU round trip on the chain, and those, not the instruction count, are what a symbol waits for
                const uint32_t ms = (uint32_t)__popcll(Mb & ((1ull << (stop & 63u)) - 1ull));
                g += PW_ADV(e);
                const uint32_t base = b.pb + g;
                D = lane >= k ? base : D;
                o = g + ms;
                zc = t + 1u;
                c = (int32_t)wl((uint32_t)PW_PAY_S(e), stop, (uint32_t)c); // ZRL: payload 0 onto a zero (stop 64: lane 0, outside every AC band)
                k = stop + 1u;
            } while (k <= Se);
            bad = bad || k == 65u || (special & PW_F_BAD);
            if (special) k = k_special;
            if (special & PW_F_EOB) { // EOBn: the rest of the band is the first block of the run
                eobrun = PW_PAY_U(special);
                const uint32_t adv = PW_ADV(special);
                o += adv;
                g += adv;
            }
        }
        if (eobrun) {
            if (k <= Se) {
                const uint32_t base = b.pb + g;
                D = lane >= k ? base : D;
                o = g + mtotal;
            }
            eobrun--;
        }
        // one correction bit for every coefficient with history (all of them: the symbols or the EOB run walked the whole band)
        const bool mine = (Mb >> lane) & 1ull;
        if (Mb) {
            const uint32_t bit = b.bit_at(D + mrank, mine);
            if (mine && bit && (c & p1) == 0) c += c >= 0 ? p1 : m1;
        }
        if (in_band) *dst = (int16_t)c;
    };
    PW_RING_LOAD_I16("v100", fetch_ptr());
    PW_RING_LOAD_I16("v101", fetch_ptr());
    PW_RING_LOAD_I16("v102", fetch_ptr());
    PW_RING_LOAD_I16("v103", fetch_ptr());
    for (uint32_t n = 0; n < nblk; n += 4u) {
        int32_t c;
        // behind the load being taken: three more loads and three stores (every block issues exactly one of each; anything else in
        // between only makes the wait stricter)
        PW_RING_TAKE(6, "v100", c);
        PW_RING_LOAD_I16("v100", fetch_ptr());
        block(c);
        if (n + 1u < nblk) { PW_RING_TAKE(6, "v101", c); PW_RING_LOAD_I16("v101", fetch_ptr()); block(c); }
        if (n + 2u < nblk) { PW_RING_TAKE(6, "v102", c); PW_RING_LOAD_I16("v102", fetch_ptr()); block(c); }
        if (n + 3u < nblk) { PW_RING_TAKE(6, "v103", c); PW_RING_LOAD_I16("v103", fetch_ptr()); block(c); }
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); // nothing of the hand-issued loads outlives the loop
    if (bad || b.pb + o > iv.end) s.irregular();
}

// ---- decode_mcu_DC_first: the scan's components interleaved, a table per component, predictors, values << Al
__device__ __forceinline__ void pw_dc_first(const PwScan& s, PwBits& b, const uint16_t* lut, const PwCanon* can)
{
    const LpProgScan& sc = *s.sc;
    const uint32_t Al = sc.Al, lane = s.lane, ns = sc.ns;
    PwIntervals iv;
    iv.init(s);
    uint32_t o = 0;
    uint32_t E0 = 0, E1 = 0, E2 = 0, E3 = 0;
    auto refill = [&](uint32_t p) __attribute__((always_inline)) {
        const uint32_t peek = b.window(p);
        o = 0;
        E0 = pw_entry_dc(pw_lookup(lut, can, PW_DC_BITS, peek), peek);
        if (ns > 1u) E1 = pw_entry_dc(pw_lookup(lut + (1u << PW_DC_BITS), can + 1, PW_DC_BITS, peek), peek);
        if (ns > 2u) E2 = pw_entry_dc(pw_lookup(lut + (2u << PW_DC_BITS), can + 2, PW_DC_BITS, peek), peek);
        if (ns > 3u) E3 = pw_entry_dc(pw_lookup(lut + (3u << PW_DC_BITS), can + 3, PW_DC_BITS, peek), peek);
    };
    b.seek(0);
    refill(0);
    int32_t pred[4] = {0, 0, 0, 0};
    bool bad = false;
    // finished blocks wait in two registers (value, block) and leave 64 at a time
    uint32_t pv = 0, pa = 0, pn = 0;
    auto flush = [&]() __attribute__((always_inline)) {
        if (lane < pn) s.coef[(size_t)pa * 64u] = (int16_t)pv;
        pn = 0;
    };
    for (uint32_t my = 0; my < sc.mcuy; my++) {
        bad = !s.wait_for(my) || bad;
        if (s.progress && my) { flush(); s.publish(my); }
        for (uint32_t mx = 0; mx < sc.mcux; mx++) {
            uint32_t np;
            if (iv.crossing(s, b.pb + o, &np, &bad)) {
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
                refill(np);
            }
            if (((b.pb + o) >> 5) - b.wb >= 32u && ((b.pb + o) >> 5) - b.wb < 61u) b.shift(false);
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                if (q >= ns) break;
                for (uint32_t v = 0; v < sc.vs[q]; v++)
                    for (uint32_t h = 0; h < sc.hs[q]; h++) {
                        if (PW_RARE(o >= 64u)) refill(b.pb + o);
                        const uint32_t Ec = q == 0u ? E0 : q == 1u ? E1 : q == 2u ? E2 : E3; // (read after the refill)
                        const uint32_t e = rl(Ec, o);
                        bad = bad || PW_SPECIAL(e);
                        o += PW_ADV(e) + (PW_SPECIAL(e) ? 1u : 0u); // (a bad code: move on, the image is the host route's anyway)
                        pred[q] += PW_PAY_S(e);
                        const uint32_t blk = sc.cblk[q] + (my * sc.vs[q] + v) * sc.bw[q] + mx * sc.hs[q] + h;
                        pv = wl((uint32_t)pred[q] << Al, pn, pv);
                        pa = wl(blk, pn, pa);
                        if (++pn == 64u) flush();
                    }
            }
        }
    }
    flush();
    if (bad || b.pb + o > iv.end) s.irregular();
}

// ---- decode_mcu_DC_refine: one raw bit per block, MCU order: lane = MCU
__device__ __forceinline__ void pw_dc_refine(const PwScan& s, const PwBits& b)
{
    const LpProgScan& sc = *s.sc;
    const uint32_t lane = s.lane, ns = sc.ns;
    const int16_t p1 = (int16_t)(1 << sc.Al);
    uint32_t bpm = 0;
    for (uint32_t q = 0; q < ns; q++) bpm += (uint32_t)sc.hs[q] * sc.vs[q];
    const uint32_t nmcu = sc.mcux * sc.mcuy, dri = sc.dri;
    // every interval must hold its MCUs' bits (else libjpeg runs into the marker: insufficient data)
    const uint32_t nint = dri ? (nmcu + dri - 1u) / dri : 1u;
    if (nint - 1u > s.n_rst) { s.irregular(); return; }
    bool bad = false;
    for (uint32_t k = lane; k < nint; k += 64u) {
        const uint32_t beg = k ? s.rst[k - 1u] : 0u, end = k < s.n_rst ? s.rst[k] : s.total_bits;
        const uint32_t m_in = dri ? (k + 1u < nint ? dri : nmcu - k * dri) : nmcu;
        bad = bad || beg + m_in * bpm > end;
    }
    if (ballot(bad)) { s.irregular(); return; }
    uint32_t waited = 0, published = 0; // MCU rows
    bool stuck = false;
    for (uint32_t m0 = 0; m0 < nmcu; m0 += 64u) {
        if (s.progress) { // the 64 MCUs of this step reach into row (m0 + 63) / mcux; rows below m0 / mcux are complete
            const uint32_t last = (m0 + 63u < nmcu ? m0 + 63u : nmcu - 1u) / sc.mcux, done = m0 / sc.mcux;
            for (; waited <= last; waited++) stuck = !s.wait_for(waited) || stuck;
            if (done > published) { s.publish(done); published = done; }
        }
        const uint32_t m = m0 + lane;
        const bool live = m < nmcu;
        const uint32_t mm = live ? m : 0u;
        const uint32_t my = mm / sc.mcux, mx = mm - my * sc.mcux;
        const uint32_t k = dri ? mm / dri : 0u;
        uint32_t bp = (k ? s.rst[k - 1u] : 0u) + (mm - k * dri) * bpm;
        for (uint32_t q = 0; q < ns; q++)
            for (uint32_t v = 0; v < sc.vs[q]; v++)
                for (uint32_t h = 0; h < sc.hs[q]; h++) {
                    const uint32_t blk = sc.cblk[q] + (my * sc.vs[q] + v) * sc.bw[q] + mx * sc.hs[q] + h;
                    const uint32_t w = live ? b.ldw(bp >> 5) : 0u;
                    if ((w >> (31u - (bp & 31u))) & 1u) s.coef[(size_t)blk * 64u] |= p1;
                    bp++;
                }
    }
    if (stuck) s.irregular();
}

} // namespace

// One wave per scan of scans[first .. first + n): one dependency level of a decode range. Sequential scans (LpProgScan::sequential: whole
// blocks, jdhuff.c decode_mcu) stay with k_prog_scan's lanes.
__global__ __launch_bounds__(64) void k_prog_wave(const LpProgScan* __restrict__ scans, uint32_t first, uint32_t n, const LpJpeg* __restrict__ streams,
                                                  LpJpegState* __restrict__ stream_states, const LpProgHuff* __restrict__ huffs,
                                                  const uint32_t* __restrict__ clean_arena, const uint32_t* __restrict__ rst_arena, int16_t* __restrict__ pcoef,
                                                  uint32_t skip_types, const LpProgDep* __restrict__ deps, uint32_t* __restrict__ progress)
{
    __shared__ uint16_t s_lut[1u << PW_AC_BITS];
    __shared__ PwCanon s_can[4];
    // Pipelined (progress != 0: every level of the range in this grid, scans sorted by level): a wave takes the next scan in line when it
    // STARTS, so that whatever scan it may have to wait for is held by a wave that is already running (workgroup ids promise no order).
    uint32_t i = blockIdx.x;
    if (progress) {
        if (threadIdx.x == 0) i = atomicAdd(progress + n, 1u); // the ticket counter sits behind the n progress words
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
    }
    if (i >= n) return;
    const LpProgScan* sc = scans + first + i;
    const LpJpeg& stream = streams[sc->stream];
    LpJpegState& st = stream_states[sc->stream];
    const bool skip = sc->sequential || ((skip_types >> ((sc->Ss ? 2u : 0u) + (sc->Ah ? 1u : 0u))) & 1u) || // development: LILLIPUT_HIP_PW_SKIP bit 0 DC first, 1 DC refine, 2 AC first, 3 AC refine
                      st.error; // (the unstuff kernels already sent the image to the host route)
    if (skip) {
        if (progress && threadIdx.x == 0) __hip_atomic_store(progress + i, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // nobody waits for this one
        return;
    }
    PwScan s;
    s.sc = sc;
    s.ht = huffs + sc->huff;
    s.coef = pcoef + sc->coef_off;
    s.rst = rst_arena + stream.rst_off;
    s.n_rst = st.n_rst;
    s.total_bits = st.clean_bytes * 8u;
    s.lane = threadIdx.x;
    s.err = &st.error;
    s.progress = progress;
    s.dep = deps + i;
    s.self = i;
    PwBits b;
    b.words = clean_arena + stream.clean_off;
    b.cap = stream.clean_cap_words;
    b.lane = threadIdx.x;
    b.wb = 0; b.cur = 0; b.pb = 0;
    if (sc->Ss == 0u) {
        if (sc->Ah == 0u) {
            for (uint32_t q = 0; q < sc->ns; q++) pw_build_lut(s_lut + (q << PW_DC_BITS), s_can + q, PW_DC_BITS, s.ht, q, threadIdx.x);
            __syncthreads();
            pw_dc_first(s, b, s_lut, s_can);
        } else
            pw_dc_refine(s, b);
    } else {
        pw_build_lut(s_lut, s_can, PW_AC_BITS, s.ht, 0, threadIdx.x);
        __syncthreads();
        if (sc->Ah == 0u) pw_ac_first(s, b, s_lut, s_can);
        else pw_ac_refine(s, b, s_lut, s_can);
    }
    s.publish(0xffffffffu);
}

void lp_launch_prog_wave(hipStream_t s, const LpProgScan* d_scans, uint32_t first, uint32_t n, const LpJpeg* d_streams, LpJpegState* d_stream_states,
                         const LpProgHuff* d_huffs, const uint32_t* d_clean, const uint32_t* d_rst, int16_t* d_pcoef, const LpProgDep* d_deps, uint32_t* d_progress)
{
    if (!n) return;
    static const uint32_t skip = getenv("LILLIPUT_HIP_PW_SKIP") ? (uint32_t)atoi(getenv("LILLIPUT_HIP_PW_SKIP")) : 0u;
    // LILLIPUT_HIP_PW_LDS_KB: unused dynamic LDS per workgroup = fewer waves per CU. Measured (profiles/r06_progressive.md): keeping the
    // waves at one per SIMD (30 KB) is SLOWER than letting the whole grid in (61 against 50 ms for 256 files of 1024 x 1024) -- the short
    // scans then queue behind the long ones they feed; 0 / 6 / 10 / 16 KB are within the noise of each other. What does matter is the
    // ticket order (critical path first, LpEngine::run_decode).
    static const uint32_t pad_kb = getenv("LILLIPUT_HIP_PW_LDS_KB") ? (uint32_t)atoi(getenv("LILLIPUT_HIP_PW_LDS_KB")) : 0u;
    hipLaunchKernelGGL(k_prog_wave, dim3(n), dim3(64), pad_kb << 10, s, d_scans, first, n, d_streams, d_stream_states, d_huffs, d_clean, d_rst, d_pcoef, skip, d_deps, d_progress);
}
