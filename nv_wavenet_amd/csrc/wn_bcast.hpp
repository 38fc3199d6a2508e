// wn_bcast.hpp -- wn::wavenet_bcast: one round of workgroups for batches between three and four tiles per CU (round 4).
//
// wavenet_wg (wn_kernels.hpp) splits the ROWS of every GEMM over the four waves of a workgroup, so the waves exchange h and x
// through LDS twice per layer and a workgroup holds at most three tiles of 16 utterances.  This organisation splits the
// UTTERANCES, four tiles per workgroup:
//
//   * every wave runs the WHOLE network for its own tile of 16 utterances.  The MFMA result tile (lane (g,j): rows 4g..4g+3 of
//     utterance j) is the next GEMM's B fragment once converted (wn_kernels.hpp, layout notes), so h and x never leave the
//     wave's registers: no activation exchange, and the four waves of a workgroup depend on each other for nothing but the
//     weights;
//   * the weights are streamed ONCE per workgroup and sample: the four per-wave streams of the packed blob (the very streams
//     wavenet_wg reads, same order) are copied global -> LDS by LDS-DMA (global_load_lds: no register round trip, 1 KiB per wave
//     instruction), each wave copying its own stream, into a ring of NSLOT stream positions per stream; every wave then reads
//     every fragment with ds_read_b128 two positions ahead of the MFMAs that use it.  NSLOT divides the layer and the head part
//     of the stream, so the LDS address of every fragment is a compile-time constant;
//   * the ring is recycled in CHUNKS of CH positions behind one bare s_barrier per chunk: before barrier k a wave makes sure
//     its own copies of chunk k+2 have landed (a counted s_waitcnt vmcnt), after it the chunk just consumed is dead for
//     everybody and is refilled, one copy per consumed position, with the stream NSLOT positions on;
//   * the compiler orders a DS read behind EVERY pending LDS-DMA of the wave (it cannot tell ring slots apart), which would
//     drain the copy queue in front of each fragment read, so the copies -- and the conditioning / dilated-tap loads of the
//     layer after next, up to E per chunk boundary -- are issued from inline assembly and waited for by hand.  VMEM returns in
//     order per wave and s_waitcnt takes an immediate, so every count is a compile-time function of what the boundaries in
//     front of it issued (BCfg::opsAt / waitAt / kWaitUse; tests/test_bcast_waits_cpu.py replays a wave's queue against these
//     tables: never too large, exact in the steady state).  Operations the counts do not know (ring stores, the compiler's own
//     few global accesses per sample) are always YOUNGER than what is waited for, which only makes a wait stricter;
//   * the running skip sum lives in the accumulator file and is touched by nothing but its MFMAs (inline assembly: the fp16
//     builds select VGPR-destination MFMAs globally, -amdgpu-mfma-vgpr-form);
//   * schedule of a layer (= the order of the stream): cur GEMM | gate arithmetic with the previous layer's skip GEMM issued
//     MFMA by MFMA between its stages | residual GEMM | bias + conditioning + dilated-tap GEMM of the next layer.
//
// Arithmetic, summation order per accumulator, rounding points and the layouts of ring / conditioning / history are those of
// wavenet_wg: samples are bit-identical and the device state is interchangeable between the organisations.
// Where its time goes (one workgroup, C3 fp16, 112 k clk per sample; LABNOTES.md round 4): 44 k is the network itself; the
// rest is the price of the weight copies (18 k), of the chunk barriers (24 k: skew between four independent waves), of the
// HBM loads and ring stores sharing the CU's vector-memory path with the copies (18 k) and of the fragment reads (4 k).
// (BTW, tiles per wave, is kept as a parameter of the code; only BTW = 1 is instantiated: a two-tile variant spilled 130
//  registers and was no faster than rounds of three-tile wavenet_wg workgroups -- removed.)
#pragma once

#include "wn_kernels.hpp"

namespace wn {

constexpr int bc_gcd(int a, int b) { return b == 0 ? a : bc_gcd(b, a % b); }
constexpr int bc_largest_divisor_le(int n, int mx) {
    int best = 1;
    for (int d = 1; d <= n && d <= mx; d++)
        if (n % d == 0) best = d;
    return best;
}

template <bool F16, int R, int S, int A, int BTW>
struct BCfg {
    using C = Cfg<F16, R, S, A, 1>;
    using P = Prec<F16>;
    static constexpr int TPF = P::TPF, EPL = P::EPL;
    static constexpr int NQ = C::NW;                       // streams of the blob = waves of a workgroup
    static constexpr int THREADS = NQ * 64;
    static constexpr int TILES_WG = NQ * BTW;
    static constexpr int RT = C::RT, ST = C::ST, AT = C::AT;
    static constexpr int KF_R = C::KF_R, KF_S = C::KF_S, KF_A = C::KF_A;
    static constexpr int FLW = C::FLW, FHWP = C::FHWP;
    static constexpr int FW_GATE = C::FW_GATE, FW_RES = C::FW_RES, FW_SKIP = C::FW_SKIP, FW_ZS = C::FW_ZS, FW_ZA = C::FW_ZA;
    static constexpr int HTW = C::HTW, STW = C::STW, ATW = C::ATW;
    // stream positions of a layer's GEMMs relative to the start of the part of the layer before (a multiple of FLW); layer 0's
    // part is short by the skip GEMM and starts at 0 (Cfg::streamPos)
    static constexpr int P_CUR = C::P_CUR, P_SKIP = C::P_SKIP, P_RES = C::P_RES, P_PREV = C::P_PREV;
    static constexpr int P0_CUR = 0, P0_RES = FW_GATE, P0_PREV = FW_GATE + FW_RES, P0_END = FLW - FW_SKIP;
    // ---- the LDS ring -------------------------------------------------------------------------------------------------
    static constexpr int RING_KB = 72;
    static constexpr int NSLOT = bc_largest_divisor_le(bc_gcd(FLW, FHWP), RING_KB / NQ);     // positions per stream
    static constexpr int RAP = 2;                          // read-ahead of the fragment FIFO, in positions
    static constexpr int CH = bc_largest_divisor_le(NSLOT, NSLOT / 4 < 4 ? NSLOT / 4 : 4);   // positions per chunk (<= 4: offset field)
    static constexpr int NCH = NSLOT / CH;
    static constexpr int E = 3;                            // conditioning / tap loads a chunk boundary issues at most
    static constexpr int REQ_LOADS = BTW * (NQ * C::COND_FR + KF_R);    // conditioning + tap fragments of one (sample, layer)
    static constexpr int REQ_GROUPS = (REQ_LOADS + E - 1) / E;
    // boundaries (= groups) inside the part of a generic layer / of layer 0
    static constexpr int GROUPS_L = FLW / CH, GROUPS_L0 = P0_END / CH;
    // loads of layer 0's request that its own groups cannot hold: issued by the last boundary of the previous sample's head
    static constexpr int REQ_HEAD = REQ_LOADS - GROUPS_L0 * E > 0 ? REQ_LOADS - GROUPS_L0 * E : 0;
    // ---- what every chunk boundary issues, and the hand-placed waits that follow from it ----------------------------------
    // The stream of a sample falls into parts: layer 0 (positions 0 .. P0_END-1), the generic layers (P_CUR .. P_CUR+FLW-1
    // each, relative to a multiple of FLW), the tail (skip GEMM of the last layer: P_CUR .. FLW-1) and the head (0 .. FHWP-1,
    // relative to the head's start).  A boundary sits behind every CH-th position (absolute numbering: BP = position + 1 is a
    // multiple of CH) and issues the next loads of the conditioning / tap request its part carries -- nothing where there is
    // nothing to request (round 4 began with dummy loads that kept every boundary at E operations: a third of the kernel's
    // vector-memory instructions).
    static constexpr int PART_L0 = 0, PART_GEN = 1, PART_TAIL = 2, PART_HEAD = 3;
    __host__ __device__ static constexpr int reqInGroup(int gi, int first) {     // loads [first + gi E, first + (gi+1) E) of a request that exist
        const int n = REQ_LOADS - first - gi * E;
        return gi < 0 || n < 0 ? 0 : n > E ? E : n;
    }
    static constexpr int BP0_GEN = (P_CUR / CH + 1) * CH;          // first boundary of a generic layer's part (and of the tail)
    __host__ __device__ static constexpr int opsAt(int part, int BP) {
        return part == PART_GEN ? reqInGroup((BP - BP0_GEN) / CH, 0)
             : part == PART_L0 ? reqInGroup(BP / CH - 1, REQ_HEAD)
             : part == PART_HEAD ? (BP == FHWP ? REQ_HEAD : 0)
                                 : 0;
    }
    // operations of the i-th boundary in front of the one at BP -- where that boundary lies in the part before, a LOWER bound
    // over the parts that can come before (a wait that allows fewer operations in flight than were issued is only stricter):
    // in front of a generic layer or the tail a generic layer is assumed (layer 0's last boundaries issue at least as much:
    // asserted below), in front of layer 0 the head, in front of the head the tail
    __host__ __device__ static constexpr int opsBefore(int part, int BP, int i) {
        const int b = BP - i * CH;
        return part == PART_GEN || part == PART_TAIL ? (b >= BP0_GEN ? opsAt(part, b) : opsAt(PART_GEN, b + FLW))
             : part == PART_L0 ? (b >= CH ? opsAt(PART_L0, b) : b == 0 ? REQ_HEAD : 0)
                               : 0;
    }
    // before barrier k: own copies of chunk k+2 landed.  Its pieces were issued one per position while chunk k+2-NCH+1 was
    // consumed (consume); behind the last of them: the loads of the boundary that ended that chunk, NCH-4 whole chunks (CH
    // pieces and their boundary's loads each) and the CH pieces of chunk k itself
    __host__ __device__ static constexpr int waitAt(int part, int BP) {
        int n = (NCH - 3) * CH;
        for (int i = 1; i <= NCH - 3; i++) n += opsBefore(part, BP, i);
        return n;
    }
    // conditioning / taps of layer l+2 are used behind the boundary in front of the tap GEMM at the end of layer l+1: the
    // youngest load of a request is followed by at least the pieces and the loads of layer l+1 up to that boundary (where
    // layer 0 is the user, by the whole head)
    static constexpr int USE_GROUPS = P_PREV / CH - P_CUR / CH;          // boundaries of a generic layer's part in front of its tap GEMM
    __host__ __device__ static constexpr int useWait() {
        int n = P_PREV - P_CUR;
        for (int gi = 0; gi < USE_GROUPS; gi++) n += reqInGroup(gi, 0);
        return n;
    }
#ifndef WN_BC_WAITU
    static constexpr int kWaitUse = useWait();
#else
    static constexpr int kWaitUse = WN_BC_WAITU;
#endif
    __host__ __device__ static constexpr bool tailsOrdered() {      // layer 0's last boundaries against a generic layer's (opsBefore)
        for (int i = 0; i < NCH - 3 && i < GROUPS_L0; i++)
            if (reqInGroup(GROUPS_L0 - 1 - i, REQ_HEAD) < reqInGroup(GROUPS_L - 1 - i, 0)) return false;
        return true;
    }
    static constexpr bool SUPPORTED =
        R == 64 && NQ == 4 && A <= 256 && S <= 256 && BTW == 1 && NSLOT >= 8 && NCH >= 4 && CH >= RAP && CH <= 4 && P_CUR % RAP == 0 &&
        FLW % RAP == 0 && FHWP % RAP == 0 && (REQ_HEAD == 0 || REQ_HEAD == E) && REQ_GROUPS <= GROUPS_L && P_PREV % CH == 0 &&
        P0_PREV % CH == 0 && FLW % CH == 0 && FHWP >= (NCH - 3) * CH && GROUPS_L0 >= NCH - 3 && tailsOrdered() &&
        (NCH - 3) * (CH + E) <= 63 && kWaitUse <= 63 && FW_SKIP + FHWP + P0_PREV >= kWaitUse;
    // gate tile of fragment-local slot `it` of stream q (Cfg: a wave's rows come in (tanh tile, sigmoid tile) pairs)
    __host__ __device__ static constexpr int gateTile(int q, int it) { return q + NQ * (it >> 1) + (it & 1) * RT; }
    // ---- LDS layout (bytes) -------------------------------------------------------------------------------------------
    static constexpr int RING_BYTES = NQ * NSLOT * 1024;
    // softmax: the lane split of wavenet_wg (LPU lanes per utterance, RPL logits per lane) -- sums of the same terms in
    // another order differ in the last bit, and a draw within that of a CDF edge would pick the neighbouring bin (measured with
    // 8 x 32 instead of 16 x 16: 31 of 4096 utterances parted from wavenet_wg's within 704 samples).  A wave takes a tile in
    // 16 / SM_U passes of SM_U utterances
    static constexpr int LPU = C::LPU, RPL = C::RPL;
    static constexpr int SM_U = 64 / LPU, SM_PASSES = 16 / SM_U;
    static_assert(LPU <= 64 && 64 % LPU == 0 && 16 % SM_U == 0, "softmax passes of a wave");
    static constexpr int LROW = A + 4;
    static constexpr int LG_BYTES = NQ * SM_U * LROW * 4;
    static constexpr int YB_BYTES = NQ * BTW * 16 * 4;
    static constexpr int OFF_RING = 0, OFF_LG = RING_BYTES, OFF_YB = OFF_LG + LG_BYTES, OFF_BIAS = OFF_YB + YB_BYTES;
    // bias table: DUMP kernels keep the per-layer running skip-bias sums (the dumps need them), production kernels only the total
    static constexpr int BIAS_LB = 3 * R;                   // gate + residual biases of a layer
    __host__ __device__ static size_t biasFloats(int L, bool dump) { return dump ? (size_t)L * C::BIAS_L + 2 * A : (size_t)L * BIAS_LB + S + 2 * A; }
    static size_t ldsBytes(int L, bool dump, int embTables) {
        return (size_t)OFF_BIAS + biasFloats(L, dump) * sizeof(float) + (size_t)embTables * A * R * sizeof(typename P::elem);
    }
};

// ---- vector memory from inline assembly (see the header: the compiler must not know these are memory operations) -----------
// Registers the compiler must keep its hands off.  A conditioning / tap load is in flight for more than a layer; if its
// destination were a compiler-visible value, the register allocator would be free to copy it (live-range splits, loop edges)
// while the load has not written it yet -- scripts/check_bcast_asm.py found exactly that.  So the two register sets are FIXED
// accumulator registers a[160:255]: the loads name them literally and have no outputs; the hand-placed wait is the statement
// that DEFINES them for the compiler (physical-register outputs, which it coalesces with the MFMA operands: no copy); and every
// assembly statement of the kernel lists all of them as clobbered, so that no value of the compiler's can live in one across
// any of these statements -- which occur every few instructions.  scripts/check_bcast_asm.py verifies on the emitted code that
// nothing but the loads touches a destination before its covering wait.
//   cd set s, fragment i (= tile*NCD + c, i < 8):  a[160 + 32 s + 4 i ...];   tap set s, fragment i (< 4):  a[224 + 16 s + 4 i ...]
#define WN_BC_R10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define WN_BC_RES                                                                                                                   \
    WN_BC_R10(16), WN_BC_R10(17), WN_BC_R10(18), WN_BC_R10(19), WN_BC_R10(20), WN_BC_R10(21), WN_BC_R10(22), WN_BC_R10(23), WN_BC_R10(24), \
        "a250", "a251", "a252", "a253", "a254", "a255", "a159"
constexpr int kBcCdReg = 160, kBcXpReg = 224;
__host__ __device__ constexpr int bc_cd_reg(int set, int i) { return kBcCdReg + 32 * set + 4 * i; }
__host__ __device__ constexpr int bc_xp_reg(int set, int i) { return kBcXpReg + 16 * set + 4 * i; }

// 1-KiB pieces of the wave's stream (at src, wave-uniform) -> LDS: lane l's 16 bytes land at M0 + 16 l, and M0 reaches the whole
// 160 KiB (scripts/ubench/ldsdma_addr.hip).
// Timing experiments (results are wrong with any of them): WN_BC_ABL_NODMA no weight copies; WN_BC_ABL_NOBAR no chunk barriers /
// waits; WN_BC_ABL_NOFIFO no fragment reads from LDS; WN_BC_ABL_NOREQ no conditioning / tap loads, no ring stores
// Piece J of a chunk: the instruction's offset field moves source AND destination (scripts/ubench/ldsdma_addr.hip), and the pieces
// of a chunk are consecutive in the stream as in the ring, so all of them share M0 (= LDS address of the chunk's first slot: the
// wave's ring base + a constant, one scalar add) and the source offset register `voff` (= lane * 16 + the chunk's position in the
// stream, advanced once per chunk).  Two instructions per piece; the first version paid six scalar ones on top of the copy.
template <int SLOT0, int J> WN_DEV void bc_dma_piece(unsigned ringMine, unsigned voff, const char* src) {
    static_assert(J >= 0 && J * 1024 < 4096, "offset field");
#ifndef WN_BC_ABL_NODMA
    asm volatile("s_add_u32 m0, %0, %1\n\tglobal_load_lds_dwordx4 %2, %3 offset:%4" ::"s"(ringMine), "n"(SLOT0 * 1024), "v"(voff), "s"(src), "n"(J * 1024)
                 : "memory", "scc", WN_BC_RES);
#endif
}
// one 16-byte-per-lane load into the fixed accumulator quad a[REG:REG+3], streaming policy; valid only behind a bc_wait_set
template <int REG> WN_DEV void bc_load_fixed(unsigned voff, rsrc_t rs, unsigned soff) {
    static_assert(REG >= kBcCdReg && REG + 3 <= 255 && REG % 4 == 0, "fixed register map");
#ifndef WN_BC_ABL_NOREQ
    asm volatile("buffer_load_dwordx4 a[%0:%1], %2, %3, %4 offen" WN_BC_LD_AUX ::"n"(REG), "n"(REG + 3), "v"(voff), "s"(rs), "s"(soff) : "memory", WN_BC_RES);
#endif
}
WN_DEV void bc_store(floatx4 v, unsigned voff, rsrc_t rs, unsigned soff) {
#if !defined(WN_BC_ABL_NOREQ) && !defined(WN_BC_ABL_NOSTORE)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" WN_BC_ST_AUX ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory", WN_BC_RES);
#endif
}
// Every load of register set SET has landed once at most N younger vector-memory operations are outstanding; from here on the
// set's registers are values the compiler may use (cd[i], i < 8; xp[i], i < 4: the first BTW*NCD / BTW*KF_R are meaningful)
template <int SET, int N> WN_DEV void bc_wait_set(floatx4 (&cd)[8], floatx4 (&xp)[4]) {
    if constexpr (SET == 0)
        asm volatile("s_waitcnt vmcnt(%12)"
                     : "={a[160:163]}"(cd[0]), "={a[164:167]}"(cd[1]), "={a[168:171]}"(cd[2]), "={a[172:175]}"(cd[3]), "={a[176:179]}"(cd[4]),
                       "={a[180:183]}"(cd[5]), "={a[184:187]}"(cd[6]), "={a[188:191]}"(cd[7]), "={a[224:227]}"(xp[0]), "={a[228:231]}"(xp[1]),
                       "={a[232:235]}"(xp[2]), "={a[236:239]}"(xp[3])
                     : "n"(N)
                     : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%12)"
                     : "={a[192:195]}"(cd[0]), "={a[196:199]}"(cd[1]), "={a[200:203]}"(cd[2]), "={a[204:207]}"(cd[3]), "={a[208:211]}"(cd[4]),
                       "={a[212:215]}"(cd[5]), "={a[216:219]}"(cd[6]), "={a[220:223]}"(cd[7]), "={a[240:243]}"(xp[0]), "={a[244:247]}"(xp[1]),
                       "={a[248:251]}"(xp[2]), "={a[252:255]}"(xp[3])
                     : "n"(N)
                     : "memory");
}
// (odd layer counts) the landed contents of set 0 become set 1's: the compiler moves the values into set 1's registers
WN_DEV void bc_set0_to_set1(const floatx4 (&cd)[8], const floatx4 (&xp)[4]) {
    asm volatile("" ::"{a[192:195]}"(cd[0]), "{a[196:199]}"(cd[1]), "{a[200:203]}"(cd[2]), "{a[204:207]}"(cd[3]), "{a[208:211]}"(cd[4]),
                 "{a[212:215]}"(cd[5]), "{a[216:219]}"(cd[6]), "{a[220:223]}"(cd[7]), "{a[240:243]}"(xp[0]), "{a[244:247]}"(xp[1]),
                 "{a[248:251]}"(xp[2]), "{a[252:255]}"(xp[3])
                 : "memory");
}

// ------------------------------------------------------------------------------------------------------------------------
// the kernel: one workgroup = NQ waves, wave w generates `count` samples for tiles tileBase + block*TILES_WG + w*BTW .. +BTW-1
// ------------------------------------------------------------------------------------------------------------------------
// EMBLDS: the current tap's embedding table is held in LDS.  DUMP: activation dump of the launch's last sample (parity mode).
template <bool F16, int R, int S, int A, int BTW, bool EMBLDS, bool DUMP>
__global__ __launch_bounds__((BCfg<F16, R, S, A, BTW>::THREADS), 1) void wavenet_bcast(const Params p) {
    using B = BCfg<F16, R, S, A, BTW>;
    using C = typename B::C;
    using P = Prec<F16>;
    using frag = typename P::frag;
    using quad = typename P::quad;
    using elem = typename P::elem;
    constexpr int NQ = B::NQ, NSLOT = B::NSLOT, CH = B::CH, RAP = B::RAP, E = B::E;
    constexpr int RT = B::RT, ST = B::ST, AT = B::AT, KF_R = B::KF_R, KF_S = B::KF_S, KF_A = B::KF_A;
    constexpr int FLW = B::FLW, FW_GATE = B::FW_GATE, FW_RES = B::FW_RES, FW_SKIP = B::FW_SKIP;
    constexpr int HTW = B::HTW, STW = B::STW, ATW = B::ATW, TPF = P::TPF;
    constexpr int COND_FR = C::COND_FR, NCD = NQ * COND_FR, REQ_HEAD = B::REQ_HEAD;
    static_assert(B::SUPPORTED, "shape not supported by wavenet_bcast");
    static_assert(RAP == 2, "FIFO priming below");
    using IC0 = std::integral_constant<int, 0>;

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const ringLds = lds + B::OFF_RING;
    float* const biasLds = (float*)(lds + B::OFF_BIAS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int L = p.numLayers;
    const int tileW = p.tileBase + blockIdx.x * B::TILES_WG + w * BTW;      // first tile of this wave
    const unsigned lane16 = (unsigned)lane * 16u;
    float* const lgMine = (float*)(lds + B::OFF_LG) + w * (B::SM_U * B::LROW);
    int* const ybMine = (int*)(lds + B::OFF_YB) + w * (BTW * 16);

    if (p.clk != nullptr && blockIdx.x == 0 && tid == 0) {
        p.clk[0] = __builtin_amdgcn_s_memtime();
        p.clk[1] = __builtin_amdgcn_s_memrealtime();
    }

    int ub[BTW];
    bool uvalid[BTW];
#pragma unroll
    for (int b = 0; b < BTW; b++) {
        const int u = (tileW + b) * 16 + j;
        uvalid[b] = u < p.batch;
        ub[b] = uvalid[b] ? u : p.batch - 1;
    }
    // softmax role of a lane: utterance su of the SM_U of a pass, logits sq*RPL .. of its row
    const int su = lane / B::LPU, sq = lane % B::LPU;

    // ---- biases -> LDS -----------------------------------------------------------------------------------------------
    // DUMP: the table as wavenet_wg keeps it ([L][Bh 2R | Bres R | running skip-bias sum S] | Bzs | Bza); otherwise
    // [L][Bh | Bres] | total skip bias | Bzs | Bza
    constexpr int BL = DUMP ? C::BIAS_L : B::BIAS_LB;
    if constexpr (DUMP) {
        const int nb = L * C::BIAS_L + 2 * A;
        for (int i = tid; i < nb; i += B::THREADS) biasLds[i] = p.bias[i];
        __syncthreads();
        for (int s0 = tid; s0 < S; s0 += B::THREADS) {
            float run = biasLds[3 * R + s0];
            for (int l = 1; l < L; l++) {
                run += biasLds[l * C::BIAS_L + 3 * R + s0];
                biasLds[l * C::BIAS_L + 3 * R + s0] = run;
            }
        }
    } else {
        for (int i = tid; i < L * B::BIAS_LB; i += B::THREADS) biasLds[i] = p.bias[(i / B::BIAS_LB) * C::BIAS_L + i % B::BIAS_LB];
        for (int s0 = tid; s0 < S; s0 += B::THREADS) {
            float run = p.bias[3 * R + s0];          // the additions in layer order, like the running sums above
            for (int l = 1; l < L; l++) run += p.bias[l * C::BIAS_L + 3 * R + s0];
            biasLds[L * B::BIAS_LB + s0] = run;
        }
        for (int i = tid; i < 2 * A; i += B::THREADS) biasLds[L * B::BIAS_LB + S + i] = p.bias[L * C::BIAS_L + i];
    }
    const float* const skipBiasTot = DUMP ? biasLds + (L - 1) * C::BIAS_L + 3 * R : biasLds + L * B::BIAS_LB;
    const float* const headBias = DUMP ? biasLds + L * C::BIAS_L : biasLds + L * B::BIAS_LB + S;

    const elem* const embPrev = (const elem*)p.embPrev;
    const elem* embCur = (const elem*)p.embCur;
    if constexpr (EMBLDS) {
        elem* const embLds = (elem*)(biasLds + B::biasFloats(L, DUMP));
        const floatx4* s0 = (const floatx4*)p.embCur;
        constexpr int CHN = (int)(A * R * sizeof(elem) / 16);
        for (int i = tid; i < CHN; i += B::THREADS) ((floatx4*)embLds)[i] = s0[i];
        embCur = embLds;
    }

    // ---- wave-uniform addressing ---------------------------------------------------------------------------------------
    const size_t strmBytes = C::waveStreamFrags(L) * 1024;                 // one stream of the blob; cyclic per sample
    const char* const wbase = (const char*)p.wblob;
    const char* const wMine = wbase + (size_t)w * strmBytes;               // this wave copies its own stream
    const unsigned ringMineLds =
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)ringLds + (unsigned)w * (unsigned)(NSLOT * 1024);
    const size_t ringTile = (size_t)p.ringSlots * KF_R * 1024;
    const rsrc_t rsRing = make_rsrc((char*)p.ring + (size_t)tileW * ringTile);
    const unsigned ringTileB = (unsigned)ringTile;
    const size_t condRow = (size_t)p.tiles * NCD * 1024;                   // one (sample, layer) row of the packed conditioning
    constexpr unsigned kCondTile = NCD * 1024;                             // bytes of one tile in a row
    // row of the request the current layer is issuing; rows are requested in the order they lie in memory
    const char* condReq = (const char*)p.cond + (size_t)tileW * kCondTile + (size_t)p.initSample * L * condRow;

    // ---- history -----------------------------------------------------------------------------------------------------
    int yPrev[BTW], yCur[BTW];
    // embedding row of the older tap: gathered one sample early where the registers allow it (one tile per wave)
    constexpr bool EP_EARLY = BTW == 1;
    floatx4 ep[BTW][RT];
#pragma unroll
    for (int b = 0; b < BTW; b++) {
        yPrev[b] = p.yInPrev[ub[b]];
        yCur[b] = p.yInCur[ub[b]];
        if constexpr (EP_EARLY) {
#pragma unroll
            for (int i = 0; i < RT; i++) ep[b][i] = quad_to_f32(*(const quad*)(embPrev + (size_t)yPrev[b] * R + i * 16 + g * 4));
        }
    }

    // ---- conditioning / dilated-tap register sets (two, by layer parity): fixed accumulator registers (see above) --------
    static_assert(BTW * NCD <= 8 && BTW * KF_R <= 4, "fixed register map of the two sets");
    // load jj of a request into set SET: jj < BTW*NCD conditioning fragment (row at `row`), else tap fragment (ring slot `slot`)
    auto req_one = [&](auto SETT, auto JJ, const char* row, unsigned slot) {
        constexpr int SET = decltype(SETT)::value, jj = decltype(JJ)::value;
#ifdef WN_BC_ABL_HOT      // (timing experiment: every request reads the same, L2-resident rows)
        row = (const char*)p.cond + (size_t)tileW * kCondTile;
        slot = 0;
#endif
#ifdef WN_BC_ABL_NOLOADS  // (timing experiment: no conditioning / tap loads, ring stores kept)
        return;
#endif
        if constexpr (jj < BTW * NCD) {
            constexpr int b = jj / NCD, c = jj % NCD;
            bc_load_fixed<bc_cd_reg(SET, jj)>(lane16 + (unsigned)(c & 3) * 1024u, make_rsrc(row), (unsigned)(b * kCondTile + (c & ~3) * 1024));
        } else {
            constexpr int i = jj - BTW * NCD, b = i / KF_R, k = i % KF_R;
            bc_load_fixed<bc_xp_reg(SET, i)>(lane16 + (unsigned)(k & 3) * 1024u, rsRing,
                                             slot * (unsigned)(KF_R * 1024) + (unsigned)b * ringTileB + (unsigned)(k & ~3) * 1024u);
        }
    };
    // the loads of a boundary's group: [J0, J0 + E) of a request, where they exist (BCfg::opsAt is the count)
    auto group_ops = [&](auto SETT, auto J0, const char* row, const unsigned slot) {
        static_for<E>([&](auto I) {
            constexpr int jj = decltype(J0)::value + decltype(I)::value;
            if constexpr (jj >= 0 && jj < B::REQ_LOADS) req_one(SETT, std::integral_constant<int, jj>{}, row, slot);
        });
    };
    using SET0 = std::integral_constant<int, 0>;
    using SET1 = std::integral_constant<int, 1>;

    // ---- the weight ring: first turn -----------------------------------------------------------------------------------
    // positions [0, NSLOT) of every stream; dmaV = lane * 16 + byte position (in the stream) of the chunk being copied
    unsigned dmaV = lane16;
    static_for<B::NCH>([&](auto CI) {
        static_for<CH>([&](auto J) { bc_dma_piece<decltype(CI)::value * CH, decltype(J)::value>(ringMineLds, dmaV, wMine); });
        dmaV += CH * 1024;
    });
    // every position consumed refills the slot CH positions back with the stream NSLOT positions on (consume): the launch's
    // first CH positions have no finished chunk behind them -- they rewrite the last CH slots with what these hold already
    dmaV = lane16 + (unsigned)((NSLOT - CH) * 1024);
    // the fragment FIFO: the NQ fragments of RAP consecutive positions, read RAP positions ahead of the MFMAs
    frag fifo[RAP][NQ];
    auto fifo_fill = [&](auto POS) {       // position POS (relative to a multiple of NSLOT) into its FIFO slot
        constexpr int pos = decltype(POS)::value;
        // (kept in this order: the uses below start with the LAST stream's fragment, so that one counted wait covers a position)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
#ifndef WN_BC_ABL_NOFIFO
            fifo[pos % RAP][q] = *(const frag*)(ringLds + ((q * NSLOT + pos % NSLOT) << 10) + lane16);
#else
            if (pos < 2) fifo[pos % RAP][q] = *(const frag*)(ringLds + ((q * NSLOT + pos % NSLOT) << 10) + lane16);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- dilation schedule: entries of layers l, l+1, l+2 travel with the layer loop (scalar arithmetic: a table in the kernel
    //      arguments would be scalar LOADS, whose out-of-order return forces lgkmcnt(0) -- a drain of the fragment FIFO)
    const Dil dS0 = dil_first();
    const Dil dS1 = dil_next(dS0, p.maxDilation, false);
    const Dil dS2 = dil_next(dS1, p.maxDilation, false);

    // ---- requests of the first sample: layers 0 and 1 in full, and the loads of layer 2's that a sample's head issues --------
    const int t0 = p.initSample;
    static_for<B::REQ_LOADS>([&](auto J) { req_one(SET0{}, J, condReq, (unsigned)(dS0.off + (t0 & (dS0.d - 1)))); });
    condReq += condRow;
    static_for<B::REQ_LOADS>([&](auto J) { req_one(SET1{}, J, condReq, (unsigned)(dS1.off + (t0 & (dS1.d - 1)))); });
    condReq += condRow;
    floatx4 cd0[8], xp0[4];                // layer 0's conditioning and tap
    bc_wait_set<0, 0>(cd0, xp0);           // (everything so far has landed, the ring's first turn included)
    __syncthreads();                       // bias table, embedding table and the ring's first turn complete for every wave

    fifo_fill(IC0{});
    fifo_fill(std::integral_constant<int, 1>{});

    // selA[tt]: A operand that copies the rows of tile tt of a B-layout fragment into a result tile (wn_kernels.hpp)
    frag selA[TPF];
#pragma unroll
    for (int tt = 0; tt < TPF; tt++)
#pragma unroll
        for (int e = 0; e < P::EPL; e++) selA[tt][e] = (elem)(((e >> 2) == tt && g * 4 + (e & 3) == j) ? 1.0f : 0.0f);

    // acc[b][T]: gate pre-activation of the next layer to run, T = 0 .. 2RT-1 (tanh rows, then sigmoid rows)
    floatx4 acc[BTW][2 * RT];
    // acc := gate bias + conditioning.  Conditioning fragment c = q*COND_FR + c' holds gate slots it = c'*TPF + tt of stream q
    auto bias_cond = [&](const float* bl, const floatx4 (&cd)[8]) {
        static_for<NCD * TPF>([&](auto CI) {
            constexpr int c = decltype(CI)::value / TPF, tt = decltype(CI)::value % TPF;
            constexpr int T = B::gateTile(c / COND_FR, (c % COND_FR) * TPF + tt);
            const floatx4 bq = *(const floatx4*)(bl + T * 16 + g * 4);
#pragma unroll
            for (int b = 0; b < BTW; b++) {
                if constexpr (F16) acc[b][T] = mma(selA[tt], __builtin_bit_cast(frag, cd[b * NCD + c]), bq);
                else acc[b][T] = bq + cd[b * NCD + c];
            }
        });
    };
    // fragment f of a GEMM with MT tile slots per stream and KF k-fragments (order of gemm_b / pack_weight_elem): k-fragment, slot
    auto frag_kf = [](int f, int MT, int KF) { const int G = MT >= 4 ? 4 : MT; return (f / G) % KF; };
    auto frag_slot = [](int f, int MT, int KF) { const int G = MT >= 4 ? 4 : MT; return (f / (G * KF)) * G + f % G; };

    // ---- layer 0 of the first sample: pre-activation formed here, the tap's weights read straight from the blob ---------
    bias_cond(biasLds, cd0);
    if (t0 >= 1) {
        static_for<FW_GATE>([&](auto FI) {
            constexpr int f = decltype(FI)::value;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const frag a = *(const frag*)(wbase + (size_t)q * strmBytes + (C::streamPos(0, C::O_PREV, L) + f) * 1024 + lane16);
                const int T = B::gateTile(q, frag_slot(f, 2 * HTW, KF_R));
#pragma unroll
                for (int b = 0; b < BTW; b++) acc[b][T] = mma(a, __builtin_bit_cast(frag, xp0[b * KF_R + frag_kf(f, 2 * HTW, KF_R)]), acc[b][T]);
            }
        });
    }

    // layer 2's request starts here, with the loads that a sample's head issues for the next sample (set A is free now)
    static_for<REQ_HEAD>([&](auto J) { req_one(SET0{}, J, condReq, (unsigned)(dS2.off + (t0 & (dS2.d - 1)))); });

    // ---- boundary of a chunk: everybody's copies of the chunk after next have landed; the chunk just consumed is refilled ----
    // The chunk just consumed is NOT refilled in one go: the four waves' copies of a chunk issued together are a burst that the
    // CU's vector-memory path takes a few hundred cycles to accept, every wave stalled on its issue (measured: 30 of 125 k cycles
    // per sample).  Its pieces follow one per position of the next chunk's consumption (consume); the counts stay what they were:
    // behind the last piece of a chunk come the E operations of the boundary that ends the chunk after, then whole groups.
    // (W = BCfg::waitAt of the boundary: the vector-memory operations issued behind the last of those copies)
    auto boundary = [&](auto W, auto&& ops) {
#if defined(WN_BC_ABL_NOWAITB)   // (timing experiment: the barrier without the wait for the copies)
        asm volatile("s_barrier" ::"n"(decltype(W)::value) : "memory", WN_BC_RES);
#elif !defined(WN_BC_ABL_NOBAR)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(decltype(W)::value) : "memory", WN_BC_RES);
#endif
        ops();
    };
    // consume stream positions [P0, P0 + N) of part PART of the sample's stream (BCfg::PART_*; positions relative to a multiple
    // of FLW, in the head to the head's start): use(f, a[]) gets the NQ fragments of position P0 + f; grp(boundary position)
    // issues the loads of every chunk boundary inside (BCfg::opsAt says how many)
    auto consume = [&](auto PARTT, auto P0T, auto NT, auto&& use, auto&& grp) {
        constexpr int PART = decltype(PARTT)::value, P0 = decltype(P0T)::value, N = decltype(NT)::value;
        static_for<N>([&](auto FI) {
            constexpr int pos = P0 + decltype(FI)::value;
            frag a[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) a[q] = fifo[pos % RAP][q];
            use(FI, a);
            fifo_fill(std::integral_constant<int, pos + RAP>{});
            // the slot CH positions back belongs to the chunk that everybody finished before the latest boundary: it receives
            // the stream NSLOT positions on, piece by piece
            constexpr int jp = pos % CH;
            bc_dma_piece<(pos - jp + NSLOT - CH) % NSLOT, jp>(ringMineLds, dmaV, wMine);
            if constexpr (jp == CH - 1) {
                // the next chunk's place in the stream, which is cyclic per sample: the copies run NSLOT - CH positions ahead
                // of the consumption and reach the end of the stream inside the head
                if constexpr (PART == B::PART_HEAD && pos + 1 + NSLOT - CH == B::FHWP) dmaV = lane16;
                else dmaV += CH * 1024;
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((pos + 1) % CH == 0) {
                boundary(std::integral_constant<int, B::waitAt(PART, pos + 1)>{}, [&]() { grp(std::integral_constant<int, pos + 1>{}); });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    auto no_loads = [&](auto) {};
    using PL0 = std::integral_constant<int, B::PART_L0>;
    using PGEN = std::integral_constant<int, B::PART_GEN>;
    using PTAIL = std::integral_constant<int, B::PART_TAIL>;
    using PHEAD = std::integral_constant<int, B::PART_HEAD>;

    // skip accumulators: the accumulator file, touched by their MFMAs only (fp16: inline assembly, see the header)
    floatx4 skip[BTW][ST];
    auto mma_skip = [&](floatx4& c, const frag a, const frag b) {
        if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : WN_BC_RES);
        else c = mma(a, b, c);
    };
    // the compiler does not know what wrote the skip accumulators: MFMA results must not be read by the VALU for a while
    auto settle_skip = [&]() {
        if constexpr (F16) {
#pragma unroll
            for (int b = 0; b < BTW; b++)
#pragma unroll
                for (int i = 0; i < ST; i += 8)
                    asm volatile("s_nop 7\n\ts_nop 7"
                                 : "+a"(skip[b][i]), "+a"(skip[b][i + 1]), "+a"(skip[b][i + 2]), "+a"(skip[b][i + 3]), "+a"(skip[b][i + 4]),
                                   "+a"(skip[b][i + 5]), "+a"(skip[b][i + 6]), "+a"(skip[b][i + 7]));
        }
    };
    static_assert(ST % 8 == 0, "settle_skip: groups of eight accumulators");
    auto to_bfrag = [&](auto KF, frag (&dst)[decltype(KF)::value], const int tile, const floatx4 v) {
        if constexpr (F16) {
#pragma unroll
            for (int r = 0; r < 4; r++) dst[tile >> 1][(tile & 1) * 4 + r] = (_Float16)v[r];
        } else {
            dst[tile] = v;
        }
    };
    using ICKR = std::integral_constant<int, KF_R>;

    const int tEnd = t0 + p.count;
    for (int t = t0; t < tEnd; t++) {
        const bool dumpNow = DUMP && p.dump && (t == tEnd - 1);
        // selectors of the utterances this lane serves in the softmax passes: pass (b, hh) -> utterance (tileW+b)*16 + hh*SM_U + su
        // (drawn in-kernel: lane q < SM_PASSES of an utterance's 16-lane row draws pass q's and the row takes it over with a row
        //  broadcast, one Philox evaluation per tile -- wavenet_wg's way of sharing the draw across its tiles)
        float selv[BTW][B::SM_PASSES];
#pragma unroll
        for (int b = 0; b < BTW; b++) {
            if (p.useRng && B::LPU == 16) {
                int sb = (tileW + b) * 16 + (sq < B::SM_PASSES ? sq : 0) * B::SM_U + su;
                sb = sb < p.batch ? sb : p.batch - 1;
                const float mine = philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)sb);
                static_for<B::SM_PASSES>([&](auto HH) { selv[b][decltype(HH)::value] = dpp_f<0x150 + decltype(HH)::value>(mine); });
            } else {
#pragma unroll
                for (int hh = 0; hh < B::SM_PASSES; hh++) {
                    int sb = (tileW + b) * 16 + hh * B::SM_U + su;
                    sb = sb < p.batch ? sb : p.batch - 1;
                    selv[b][hh] = p.useRng ? philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)sb) : p.sel[(size_t)t * p.maxBatch + sb];
                }
            }
        }

        // ---- embedding (nv_wavenet_reference.cpp:42-56) ------------------------------------------------------------------
        floatx4 x[BTW][RT];
        frag xb[BTW][KF_R];
#pragma unroll
        for (int b = 0; b < BTW; b++) {
#pragma unroll
            for (int i = 0; i < RT; i++) {
                const floatx4 ec = quad_to_f32(*(const quad*)(embCur + (size_t)yCur[b] * R + i * 16 + g * 4));
                floatx4 epv;
                if constexpr (EP_EARLY) epv = ep[b][i];
                else epv = quad_to_f32(*(const quad*)(embPrev + (size_t)yPrev[b] * R + i * 16 + g * 4));
                floatx4 v = epv + ec;
                if (p.tanhEmbed) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = tanh_t<F16>(v[r]);
                }
                x[b][i] = v;
                to_bfrag(ICKR{}, xb[b], i, v);
            }
            if constexpr (EP_EARLY) {
#pragma unroll
                for (int i = 0; i < RT; i++) ep[b][i] = quad_to_f32(*(const quad*)(embPrev + (size_t)yCur[b] * R + i * 16 + g * 4));
            }
        }
#pragma unroll
        for (int b = 0; b < BTW; b++)
#pragma unroll
            for (int i = 0; i < ST; i++) skip[b][i] = floatx4{0.f, 0.f, 0.f, 0.f};

        frag hbA[BTW][KF_R], hbB[BTW][KF_R];     // h of the even / odd layers as B fragments

        // ---- one layer.  SK: the previous layer's skip GEMM runs under this layer's gate (every layer but layer 0) ---------
        // setC: register set of this layer's parity, free (consumed at the end of the previous layer): receives the request of
        // layer l+2; the other set holds conditioning and tap of layer l+1.  hbP: h of the previous layer, hbC: receives this layer's
        auto layer = [&](auto withSkip, auto setC, const int l, const Dil dl, const Dil dN, const Dil dl2, const frag (&hbP)[BTW][KF_R],
                         frag (&hbC)[BTW][KF_R]) {
            constexpr bool SK = decltype(withSkip)::value;
            constexpr int SETC = decltype(setC)::value, SETN = 1 - SETC;
            constexpr int PC = SK ? B::P_CUR : B::P0_CUR, PS = B::P_SKIP, PR = SK ? B::P_RES : B::P0_RES, PP = SK ? B::P_PREV : B::P0_PREV;
            const float* bl = biasLds + l * BL;
            const float* blN = biasLds + (l + 1 < L ? l + 1 : 0) * BL;
            const bool havePrevN = (l + 1 < L ? t : t + 1) >= dN.d;
            // the request of layer l+2 (of the next sample past the last layer): its ring slot
            const unsigned slot2 = (unsigned)(dl2.off + ((l + 2 >= L ? t + 1 : t) & (dl2.d - 1)));
            // group of boundary position BP: the first groups of a generic layer's part carry its request; layer 0's carry the
            // loads from REQ_HEAD on (the previous sample's head issued the first REQ_HEAD)
            auto grp = [&](auto BP) {
                constexpr int gi = SK ? (decltype(BP)::value - B::BP0_GEN) / CH : decltype(BP)::value / CH - 1;
                group_ops(setC, std::integral_constant<int, SK ? gi * E : REQ_HEAD + gi * E>{}, condReq, slot2);
            };
            using PART = std::conditional_t<SK, PGEN, PL0>;

            // -- current tap on top of bias + conditioning + dilated tap --
            consume(PART{}, std::integral_constant<int, PC>{}, std::integral_constant<int, FW_GATE>{},
                    [&](auto FI, const frag (&a)[NQ]) {
                        constexpr int f = decltype(FI)::value;
#pragma unroll
                        for (int q = NQ - 1; q >= 0; q--) {
                            const int T = B::gateTile(q, frag_slot(f, 2 * HTW, KF_R));
#pragma unroll
                            for (int b = 0; b < BTW; b++) acc[b][T] = mma(a[q], xb[b][frag_kf(f, 2 * HTW, KF_R)], acc[b][T]);
                        }
                    },
                    grp);
            // x_l[t] replaces x_l[t-d] in the ring (operations the hand-placed counts do not know: younger than anything that
            // is waited for)
            {
                const unsigned rp = (unsigned)(dl.off + (t & (dl.d - 1))) * (unsigned)(KF_R * 1024);
#pragma unroll
                for (int b = 0; b < BTW; b++)
#pragma unroll
                    for (int k = 0; k < KF_R; k++)
                        bc_store(__builtin_bit_cast(floatx4, xb[b][k]), lane16 + (unsigned)(k & 3) * 1024u, rsRing,
                                 rp + (unsigned)b * ringTileB + (unsigned)(k & ~3) * 1024u);
            }
            __builtin_amdgcn_sched_barrier(0);

            // -- gate in stages (wn_kernels.hpp: gate_stage), the previous layer's skip GEMM between the stages --
            {
                constexpr int NS = BTW * RT * 2 * 5;                      // gate stages
                constexpr int NM = FW_SKIP * NQ * BTW;                    // MFMA slots of the skip GEMM
                floatx2 ea, eb, ra, rb, hp;
                floatx4 hv;
                auto stage = [&](auto SI) {
                    constexpr int s = decltype(SI)::value, pr = s / 5, st = s % 5;
                    constexpr int b = pr / (2 * RT), i = (pr / 2) % RT, r = (pr & 1) * 2;
                    gate_stage<F16, st>(acc[b][i][r], acc[b][i][r + 1], acc[b][i + RT][r], acc[b][i + RT][r + 1], ea, eb, ra, rb, hp);
                    if constexpr (st == 4) {
                        hv[r] = hp[0];
                        hv[r + 1] = hp[1];
                        if constexpr (r == 2) to_bfrag(ICKR{}, hbC[b], i, hv);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                if constexpr (SK) {
                    consume(PART{}, std::integral_constant<int, PS>{}, std::integral_constant<int, FW_SKIP>{},
                            [&](auto FI, const frag (&a)[NQ]) {
                                constexpr int f = decltype(FI)::value;
                                static_for<NQ * BTW>([&](auto MI) {
                                    constexpr int q = NQ - 1 - decltype(MI)::value / BTW, b = decltype(MI)::value % BTW;
                                    constexpr int m = f * NQ * BTW + decltype(MI)::value;
                                    constexpr int G = STW >= 4 ? 4 : STW, kf = (f / G) % KF_R, T = q + NQ * ((f / (G * KF_R)) * G + f % G);
                                    mma_skip(skip[b][T], a[q], hbP[b][kf]);
                                    __builtin_amdgcn_sched_barrier(0);
                                    static_for_range<m * NS / NM, (m + 1) * NS / NM>(stage);
                                });
                            },
                            grp);
                    if (dumpNow) {
                        settle_skip();
                        const float* bp = biasLds + (l - 1) * C::BIAS_L + 3 * R;   // running bias sum (DUMP table)
#pragma unroll
                        for (int b = 0; b < BTW; b++) {
                            if (!uvalid[b]) continue;
#pragma unroll
                            for (int i = 0; i < ST; i++)
                                *(floatx4*)(p.skipOut + ((size_t)(l - 1) * p.maxBatch + ub[b]) * S + i * 16 + g * 4) =
                                    skip[b][i] + *(const floatx4*)(bp + i * 16 + g * 4);
                        }
                    }
                } else {
                    static_for<NS>(stage);
                }
            }

            // -- residual: x <- Wres h + Bres + x --
            floatx4 xa[BTW][RT];
#pragma unroll
            for (int i = 0; i < RT; i++) {
                const floatx4 bq = *(const floatx4*)(bl + 2 * R + i * 16 + g * 4);
#pragma unroll
                for (int b = 0; b < BTW; b++) xa[b][i] = bq + x[b][i];
            }
            consume(PART{}, std::integral_constant<int, PR>{}, std::integral_constant<int, FW_RES>{},
                    [&](auto FI, const frag (&a)[NQ]) {
                        constexpr int f = decltype(FI)::value;
#pragma unroll
                        for (int q = NQ - 1; q >= 0; q--) {
                            const int T = q + NQ * frag_slot(f, HTW, KF_R);
#pragma unroll
                            for (int b = 0; b < BTW; b++) xa[b][T] = mma(a[q], hbC[b][frag_kf(f, HTW, KF_R)], xa[b][T]);
                        }
                    },
                    grp);

            // -- the next layer's pre-activation: bias + conditioning + dilated tap (layer 0 of the next sample after the last layer) --
            floatx4 cdN[8], xpN[4];
            bc_wait_set<SETN, B::kWaitUse>(cdN, xpN);
            bias_cond(blN, cdN);
            consume(PART{}, std::integral_constant<int, PP>{}, std::integral_constant<int, FW_GATE>{},
                    [&](auto FI, const frag (&a)[NQ]) {
                        constexpr int f = decltype(FI)::value;
                        if (havePrevN) {       // (before the start the tap is zero, reference :287)
#pragma unroll
                            for (int q = NQ - 1; q >= 0; q--) {
                                const int T = B::gateTile(q, frag_slot(f, 2 * HTW, KF_R));
#pragma unroll
                                for (int b = 0; b < BTW; b++)
                                    acc[b][T] = mma(a[q], __builtin_bit_cast(frag, xpN[b * KF_R + frag_kf(f, 2 * HTW, KF_R)]), acc[b][T]);
                            }
                        }
                    },
                    grp);
            // x of the next layer, as fp32 residual stream and as B fragments
#pragma unroll
            for (int b = 0; b < BTW; b++)
#pragma unroll
                for (int i = 0; i < RT; i++) {
                    x[b][i] = xa[b][i];
                    to_bfrag(ICKR{}, xb[b], i, xa[b][i]);
                }
            if (dumpNow) {
#pragma unroll
                for (int b = 0; b < BTW; b++) {
                    if (!uvalid[b]) continue;
#pragma unroll
                    for (int i = 0; i < RT; i++) *(floatx4*)(p.xtOut + ((size_t)l * p.maxBatch + ub[b]) * R + i * 16 + g * 4) = x[b][i];
                }
            }
            condReq += condRow;                    // the next layer requests the next row
            __builtin_amdgcn_sched_barrier(0);
        };

        {
            Dil da = dS0, db = dS1, dc = dS2;      // schedule entries of layers l, l+1, l+2 (layers 0, 1 of the next sample past the end)
            auto advance = [&](int l) {            // -> entries of l+1, l+2, l+3
                da = db;
                db = dc;
                dc = dil_next(dc, p.maxDilation, l + 3 == L);
            };
            layer(std::false_type{}, SET0{}, 0, da, db, dc, hbB, hbA);
            advance(0);
            int l = 1;
            for (; l + 1 < L; l += 2) {
                layer(std::true_type{}, SET1{}, l, da, db, dc, hbA, hbB);
                advance(l);
                layer(std::true_type{}, SET0{}, l + 1, da, db, dc, hbB, hbA);
                advance(l + 1);
            }
            if (l < L) {
                layer(std::true_type{}, SET1{}, l, da, db, dc, hbA, hbB);
            } else {
                // odd layer count: the last layer was an even one.  Its h sits in the even set, and the next sample's layer 1 was
                // requested into register set 0 while layer 1 reads set 1: once the request has landed its contents move over
                floatx4 cdT[8], xpT[4];
                bc_wait_set<0, 0>(cdT, xpT);
                bc_set0_to_set1(cdT, xpT);
#pragma unroll
                for (int b = 0; b < BTW; b++)
#pragma unroll
                    for (int k = 0; k < KF_R; k++) hbB[b][k] = hbA[b][k];
            }
        }

        // ---- skip GEMM of the last layer, then the output head (nv_wavenet_reference.cpp:94-104) ----------------------------
        // the last boundary of the head issues the first REQ_HEAD loads of layer 0's request of the next sample (layer 2)
        const unsigned slotH = (unsigned)(dS2.off + ((t + 1) & (dS2.d - 1)));
        auto grpHead = [&](auto BP) {
            if constexpr (decltype(BP)::value == B::FHWP && REQ_HEAD > 0) group_ops(SET0{}, IC0{}, condReq, slotH);
        };
        consume(PTAIL{}, std::integral_constant<int, B::P_CUR>{}, std::integral_constant<int, FW_SKIP>{},
                [&](auto FI, const frag (&a)[NQ]) {
                    constexpr int f = decltype(FI)::value;
                    constexpr int G = STW >= 4 ? 4 : STW, kf = (f / G) % KF_R, sl = (f / (G * KF_R)) * G + f % G;
#pragma unroll
                    for (int q = NQ - 1; q >= 0; q--)
#pragma unroll
                        for (int b = 0; b < BTW; b++) mma_skip(skip[b][q + NQ * sl], a[q], hbB[b][kf]);
                },
                no_loads);
        settle_skip();
        floatx4 zs[BTW][AT];
        {
            frag sb[BTW][KF_S];
#pragma unroll
            for (int b = 0; b < BTW; b++)
#pragma unroll
                for (int i = 0; i < ST; i++) {
                    floatx4 v = skip[b][i] + *(const floatx4*)(skipBiasTot + i * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = __builtin_fmaxf(v[r], 0.f);
                    to_bfrag(std::integral_constant<int, KF_S>{}, sb[b], i, v);
                    if (dumpNow && uvalid[b])   // the oracle applies the ReLU to the last layer's skipOut in place
                        *(floatx4*)(p.skipOut + ((size_t)(L - 1) * p.maxBatch + ub[b]) * S + i * 16 + g * 4) = v;
                }
#pragma unroll
            for (int i = 0; i < AT; i++) {
                const floatx4 bq = *(const floatx4*)(headBias + i * 16 + g * 4);
#pragma unroll
                for (int b = 0; b < BTW; b++) zs[b][i] = bq;
            }
            consume(PHEAD{}, IC0{}, std::integral_constant<int, B::FW_ZS>{},
                    [&](auto FI, const frag (&a)[NQ]) {
                        constexpr int f = decltype(FI)::value;
#pragma unroll
                        for (int q = NQ - 1; q >= 0; q--) {
                            const int T = q + NQ * frag_slot(f, ATW, KF_S);
#pragma unroll
                            for (int b = 0; b < BTW; b++) zs[b][T] = mma(a[q], sb[b][frag_kf(f, ATW, KF_S)], zs[b][T]);
                        }
                    },
                    grpHead);
            consume(PHEAD{}, std::integral_constant<int, B::FW_ZS>{}, std::integral_constant<int, C::PAD1>{}, [&](auto, const frag (&)[NQ]) {}, grpHead);
        }
        floatx4 za[BTW][AT];
        {
            frag zb[BTW][KF_A];
#pragma unroll
            for (int b = 0; b < BTW; b++)
#pragma unroll
                for (int i = 0; i < AT; i++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) zs[b][i][r] = __builtin_fmaxf(zs[b][i][r], 0.f);
                    to_bfrag(std::integral_constant<int, KF_A>{}, zb[b], i, zs[b][i]);
                    if (dumpNow && uvalid[b]) *(floatx4*)(p.zs + (size_t)ub[b] * A + i * 16 + g * 4) = zs[b][i];
                }
#pragma unroll
            for (int i = 0; i < AT; i++) {
                const floatx4 bq = *(const floatx4*)(headBias + A + i * 16 + g * 4);
#pragma unroll
                for (int b = 0; b < BTW; b++) za[b][i] = bq;
            }
            consume(PHEAD{}, std::integral_constant<int, C::O_ZA>{}, std::integral_constant<int, B::FW_ZA>{},
                    [&](auto FI, const frag (&a)[NQ]) {
                        constexpr int f = decltype(FI)::value;
#pragma unroll
                        for (int q = NQ - 1; q >= 0; q--) {
                            const int T = q + NQ * frag_slot(f, ATW, KF_A);
#pragma unroll
                            for (int b = 0; b < BTW; b++) za[b][T] = mma(a[q], zb[b][frag_kf(f, ATW, KF_A)], za[b][T]);
                        }
                    },
                    grpHead);
            consume(PHEAD{}, std::integral_constant<int, C::O_ZA + B::FW_ZA>{}, std::integral_constant<int, C::PAD2>{}, [&](auto, const frag (&)[NQ]) {},
                    grpHead);
        }

        // ---- softmax + inverse-CDF pick: SM_U utterances per pass through the wave's own logits rows, LPU lanes per utterance ----
        // (DS operations of one wave execute in order: no barrier between a pass's row writes and its reads)
#pragma unroll
        for (int b = 0; b < BTW; b++) {
            if (dumpNow && uvalid[b]) {
#pragma unroll
                for (int i = 0; i < AT; i++) *(floatx4*)(p.za + (size_t)ub[b] * A + i * 16 + g * 4) = za[b][i];
            }
#pragma unroll
            for (int hh = 0; hh < B::SM_PASSES; hh++) {
                if (j / B::SM_U == hh) {
#pragma unroll
                    for (int i = 0; i < AT; i++) *(floatx4*)(lgMine + (j % B::SM_U) * B::LROW + i * 16 + g * 4) = za[b][i];
                }
                float e[B::RPL];
                float total;
                const int pick = softmax_pick<A, B::LPU, B::RPL>(lgMine + su * B::LROW + sq * B::RPL, sq, lane, selv[b][hh], e, total);
                const int sbu = (tileW + b) * 16 + hh * B::SM_U + su;
                if (sq == 0) {
                    ybMine[b * 16 + hh * B::SM_U + su] = pick;
                    if (sbu < p.batch) p.yOut[(size_t)sbu * p.numSamples + t] = pick;
                }
                if (dumpNow && sbu < p.batch) {
                    const float inv = 1.0f / total;
#pragma unroll
                    for (int i = 0; i < B::RPL / 4; i++)
                        *(floatx4*)(p.p + (size_t)sbu * A + sq * B::RPL + i * 4) =
                            floatx4{e[i * 4] * inv, e[i * 4 + 1] * inv, e[i * 4 + 2] * inv, e[i * 4 + 3] * inv};
                }
            }
        }
#pragma unroll
        for (int b = 0; b < BTW; b++) {
            yPrev[b] = yCur[b];
            yCur[b] = ybMine[b * 16 + j];
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing of this wave is still on its way into LDS or registers
    if (g == 0) {
#pragma unroll
        for (int b = 0; b < BTW; b++)
            if (uvalid[b]) {
                p.yInPrev[ub[b]] = yPrev[b];
                p.yInCur[ub[b]] = yCur[b];
            }
    }
    if (p.clk != nullptr && blockIdx.x == 0 && tid == 0) {
        p.clk[2] = __builtin_amdgcn_s_memtime();
        p.clk[3] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace wn
