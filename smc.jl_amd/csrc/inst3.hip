// inst3.hip - one (n_para, proposal kind, riding?, chunks per worker) instantiation of the persistent segment kernel (compile with
// -DSMCMI_INST3_D=<1..10> -DSMCMI_INST3_A=<1: α = 1, 0: mixture> -DSMCMI_INST3_R=<0: two hand-overs per stage, 1: one - fixed schedules, stage3.hpp
// k3_rides> [-DSMCMI_INST3_C=2: two 512-particle chunks per worker, α = 1 only]; see launch2.hpp).
#if !defined(SMCMI_INST3_D) || !defined(SMCMI_INST3_A) || !defined(SMCMI_INST3_R)
#error "compile with -DSMCMI_INST3_D=<n_para> -DSMCMI_INST3_A=<0|1> -DSMCMI_INST3_R=<0|1> [-DSMCMI_INST3_C=2]"
#endif
#ifndef SMCMI_INST3_C
#define SMCMI_INST3_C 1
#endif
#ifndef SMCMI_INST3_S
#define SMCMI_INST3_S 0               // 1: several handles (Seg3Args::peers)
#endif
#if SMCMI_INST3_C == 2
#define SMCMI_K3_CH2 1              // (stage3.hpp: the two-chunk text of the segment kernel)
#endif
#define SMCMI_INST_UNIT 1               // (the engines' non-template kernels belong to smcmi.hip: kernels.hpp)
#include "launch2.hpp"
