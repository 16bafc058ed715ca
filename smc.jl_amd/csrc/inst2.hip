// inst2.hip - one n_para's instantiations of the engine 2 kernels (compile with -DSMCMI_INST_D=<1..16>; see launch2.hpp).
#ifndef SMCMI_INST_D
#error "compile with -DSMCMI_INST_D=<n_para>"
#endif
#define SMCMI_INST_UNIT 1               // (the engines' non-template kernels belong to smcmi.hip: kernels.hpp)
#include "launch2.hpp"
