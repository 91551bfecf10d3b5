"""mira_amd: MI355X (gfx950) MSM + NTT engine behind joshbeal/mira's `CommitmentKey::commit`
(src/commitment.rs) and `fft` (src/fft.rs).  HIP only: importing the compute modules needs
mira_amd/csrc/libmira_gpu.so (build it with `__graft_entry__.build()`)."""
from .commitment import CURVE_BN256, CURVE_GRUMPKIN, CommitmentKey, TooLongInput, combine_partials, concatenate_with_padding  # noqa: F401
