// inst2.hip - one n_para's instantiations of the engine 2 kernels (compile with -DSMCMI_INST_D=<1..16>; see launch2.hpp).
#ifndef SMCMI_INST_D
#error "compile with -DSMCMI_INST_D=<n_para>"
#endif
#include "launch2.hpp"
