// hipadj_tu_family.hip — translation units of the workgroup-per-trajectory (Brusselator) and FP64-MFMA (MLP) families:
//   hipcc -DHIPADJ_TU_FIELD=32     Brusselator grid 8 / 16 / 32
//   hipcc -DHIPADJ_TU_MLP=128      hidden width 32 / 64 / 128
#include "hipadj_host_impl.hpp"

#if defined(HIPADJ_TU_FIELD)
template int field_forward<HIPADJ_TU_FIELD>(hipadj_handle*, const double*, const double*, double*);
template int field_adjoint<HIPADJ_TU_FIELD>(hipadj_handle*, const double*, double*, double*);
#elif defined(HIPADJ_TU_MLP)
template int mlp_forward_launch<HIPADJ_TU_MLP>(hipadj_handle*, const double*, const double*, double*);
template int mlp_adjoint_launch<HIPADJ_TU_MLP>(hipadj_handle*, const double*, double*, double*);
#else
#error "define HIPADJ_TU_FIELD or HIPADJ_TU_MLP"
#endif
