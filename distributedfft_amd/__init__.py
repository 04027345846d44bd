"""distributedfft_amd -- MI355X-native distributed 3-D FFT behind the plan/execute interface of
eggersn/DistributedFFT.  See DESIGN.md and include/dfft_c.h."""
from ._lib import DfftError, LIB_PATH  # noqa: F401
from .api import (All2All, Comm, Configurations, DeviceBuffer, FORWARD, GlobalSize, INVERSE, MPI_Type, MPIcuFFT,  # noqa: F401
                  MPIcuFFT_Pencil, MPIcuFFT_Pencil_Opt1, MPIcuFFT_Slab, MPIcuFFT_Slab_Opt1, MPIcuFFT_Slab_Y_Then_ZX, MPIcuFFT_Slab_Z_Then_YX,
                  MPIcuFFT_Slab_Z_Then_YX_Opt1, Partition, Partition_Dimensions,
                  Peer2Peer, Pencil_Partition, Slab_Partition, Streams, Sync, axis_plan_info, fft1d_batched, kernel_info, last_placement_info)
