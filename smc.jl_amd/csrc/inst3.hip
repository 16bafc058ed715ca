// inst3.hip - one n_para's instantiation of the persistent segment kernel (compile with -DSMCMI_INST3_D=<1..16>: empty beyond 10; see launch2.hpp).
#ifndef SMCMI_INST3_D
#error "compile with -DSMCMI_INST3_D=<n_para>"
#endif
#include "launch2.hpp"
