// hipadj_tu_lane.hip — one translation unit per compiled-in lane-per-trajectory model and stepper:
//   hipcc -DHIPADJ_TU_MODEL=ModelLorenz -DHIPADJ_TU_PART=0   fixed-step RK4 kernels (forward_impl, adjoint_impl)
//   hipcc -DHIPADJ_TU_MODEL=ModelLorenz -DHIPADJ_TU_PART=1   adaptive Tsit5 kernels (adaptive_forward, adaptive_adjoint)
// build.py compiles the units in parallel; hipadj_api.hip only sees the declarations (hipadj_host.hpp).
#include "hipadj_host_impl.hpp"

#if HIPADJ_TU_PART == 0
template int forward_impl<HIPADJ_TU_MODEL>(hipadj_handle*, const double*, const double*, double*);
template int adjoint_impl<HIPADJ_TU_MODEL>(hipadj_handle*, const double*, double*, double*);
#else
template int adaptive_forward<HIPADJ_TU_MODEL>(hipadj_handle*, const double*, const double*, double*);
template int adaptive_adjoint<HIPADJ_TU_MODEL>(hipadj_handle*, const double*, double*, double*);
#endif
