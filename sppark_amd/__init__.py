"""sppark_amd — MI355X-native MSM + NTT primitives behind sppark's C ABI.

Host-side mirror of the reference's Rust/Go wrappers (poc/msm-cuda/src/lib.rs,
poc/ntt-cuda/src/lib.rs, rust/src/lib.rs) over the C-ABI libraries in
sppark_amd/lib (declared in include/sppark_amd.h).  All compute happens in the
HIP libraries; importing this package without them built raises at first use.
"""
from .ffi import SpparkError, load, lib_path, cuda_available                      # noqa: F401
from .msm import (multi_scalar_mult, multi_scalar_mult_arkworks, multi_scalar_mult_fp2_arkworks, MsmContext,
                  jacobian_sum_g2, to_affine_g2, set_g2_path,      # noqa: F401
                  jacobian_sum, to_affine, generate_points, generate_progression, ngpus, msm_multi, msm_multi_shards, batch_addition)
from .ntt import (NTT, iNTT, coset_NTT, coset_iNTT, compute_ntt, LDE, LDE_powers, LDE_expand,                  # noqa: F401
                  NTTInputOutputOrder, NTTDirection, NTTType)
from . import poly                                                                   # noqa: F401
