// occupancy.hip -- resident workgroups per CU the runtime predicts for the tick kernels.
#include "../../quadrotor_landing_amd/csrc/ekf_quad_kernels.hpp"
#include <cstdio>
using namespace qle;
template <typename K> static void show(const char* name, K k, int block)
{
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, block, 0);
    hipFuncAttributes a;
    hipError_t e2 = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k));
    printf("%-40s blocks/CU=%d (err %d) regs=%d lds=%zu scratch=%zu (err %d)\n", name, n, (int)e, a.numRegs, a.sharedSizeBytes, a.localSizeBytes, (int)e2);
}
int main()
{
    show("kw_tick<f32,predict,nt2>", kw_tick<float, false, false, false, false, 2>, 256);
    show("kw_tick<f32,step,nt2>", kw_tick<float, true, false, false, true, 2>, 256);
    show("kw_tick<f64,predict,nt0>", kw_tick<double, false, false, false, false, 0>, 256);
    show("kw_tick<f64,step,nt0>", kw_tick<double, true, false, false, true, 0>, 256);
    show("k_predict<f32,nt2>", k_predict<float, false, 2, false>, 256);
    show("k_step<f32,nt2>", k_step<float, true, false, false, 2>, 256);
    return 0;
}
