// prt_kernels_f32.hip — the fp32 fast mode (PRT_PRECISION_F32): K1 and K3 of prt_kernels.hip compiled with every real
// number a float (48-byte triangle and shading records, half the registers of the fp64 path, hardware rcp / rsq / sqrt /
// exp2 / log2 / sin / cos).  Same traversal, same box test, same scheduling, same random streams; results agree with
// the fp64 path within the tier-2 tolerance, not bit for bit.  Launchers live in namespace prt32.
#define PRT_REAL float
#define PRT_F32_TU 1
#include "prt_kernels.hip"
