// inst3.hip - one (n_para, proposal kind) instantiation of the persistent segment kernel (compile with -DSMCMI_INST3_D=<1..16>
// -DSMCMI_INST3_A=<1: α = 1, 0: mixture>: empty beyond n_para 10; see launch2.hpp).
#if !defined(SMCMI_INST3_D) || !defined(SMCMI_INST3_A)
#error "compile with -DSMCMI_INST3_D=<n_para> -DSMCMI_INST3_A=<0|1>"
#endif
#include "launch2.hpp"
