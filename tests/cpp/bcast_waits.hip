// Test helper (CPU): prints the constants and the per-boundary tables of wn::BCfg's hand-placed-wait model (wn_bcast.hpp) for one
// shape, for tests/test_bcast_waits_cpu.py to replay against a model of the wave's in-order vector-memory queue.
//   bcast_waits <fp16: 0|1>     (shape from -DWN_R / -DWN_S / -DWN_A at compile time)
#include <cstdio>
#include <cstdlib>

#include "wn_bcast.hpp"

template <bool F16> static void dump() {
    using B = wn::BCfg<F16, WN_R, WN_S, WN_A, 1>;
    printf("supported %d\n", (int)B::SUPPORTED);
    if (!B::SUPPORTED) return;
    printf("const NSLOT %d CH %d NCH %d E %d REQ_LOADS %d REQ_HEAD %d FLW %d FHWP %d P_CUR %d P_PREV %d P0_PREV %d P0_END %d FW_SKIP %d kWaitUse %d RAP %d\n",
           B::NSLOT, B::CH, B::NCH, B::E, B::REQ_LOADS, B::REQ_HEAD, B::FLW, B::FHWP, B::P_CUR, B::P_PREV, B::P0_PREV, B::P0_END, B::FW_SKIP,
           B::kWaitUse, B::RAP);
    struct Part { int id; const char* name; int first, last; };
    const Part parts[] = {{B::PART_L0, "L0", 0, B::P0_END}, {B::PART_GEN, "GEN", B::P_CUR, B::P_CUR + B::FLW}, {B::PART_TAIL, "TAIL", B::P_CUR, B::FLW},
                          {B::PART_HEAD, "HEAD", 0, B::FHWP}};
    for (const Part& p : parts)
        for (int bp = (p.first / B::CH + 1) * B::CH; bp <= p.last; bp += B::CH)
            printf("boundary %s %d ops %d wait %d\n", p.name, bp, B::opsAt(p.id, bp), B::waitAt(p.id, bp));
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1])) dump<true>();
    else dump<false>();
    return 0;
}
