// inst2b.hip - one n_para's instantiation of the large-shard mutation kernel (stage2b.hpp; compile with -DSMCMI_INST2B_D=<1..10> and the
// Makefile's BIGFLAGS: without machine LICM the kernel keeps 4 wavefronts per SIMD; see launch2.hpp).
#ifndef SMCMI_INST2B_D
#error "compile with -DSMCMI_INST2B_D=<n_para>"
#endif
#define SMCMI_INST_UNIT 1               // (the engines' non-template kernels belong to smcmi.hip: kernels.hpp)
#include "launch2.hpp"
