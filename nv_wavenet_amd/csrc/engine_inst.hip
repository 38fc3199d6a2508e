// engine_inst.hip -- one nvWavenetInfer instantiation per object file:
//   hipcc -DWN_R=64 -DWN_S=256 -DWN_A=256 -DWN_P=16 -c engine_inst.hip -o inst_64_256_256_p16.o
#include "engine_base.hpp"

#if WN_P == 16
typedef half2 Tw;
typedef half Td;
#else
typedef float Tw;
typedef float Td;
#endif

nvw_engine* WN_FACTORY_NAME(WN_R, WN_S, WN_A, WN_P)(int L, int maxD, int B, int N, int impl, int tanhEmbed, int organisation) {
    return new EngineImpl<Tw, Td, WN_R, WN_S, WN_A>(L, maxD, B, N, impl, tanhEmbed != 0, organisation);
}
