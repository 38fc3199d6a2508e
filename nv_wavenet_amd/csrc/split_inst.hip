// split_inst.hip -- wn::wavenet_split for one shape (fp16):
//   hipcc -DWN_R=64 -DWN_S=256 -DWN_A=256 -c split_inst.hip -o split_64_256_256.o
// One, two or three tiles per workgroup; with / without the current tap's embedding table in LDS; conditioning packed
// (RAW = 0) or read in place from the caller's fp16 tensor (RAW = 2).
#include "wn_split.hpp"

#define WN_SPLIT_INST(BT, EMB, RAW) template __global__ void wn::wavenet_split<WN_R, WN_S, WN_A, BT, EMB, RAW>(const wn::Params);
#define WN_SPLIT_INST_BT(BT) WN_SPLIT_INST(BT, false, 0) WN_SPLIT_INST(BT, true, 0) WN_SPLIT_INST(BT, false, 2) WN_SPLIT_INST(BT, true, 2)
WN_SPLIT_INST_BT(1)
WN_SPLIT_INST_BT(2)
WN_SPLIT_INST_BT(3)
