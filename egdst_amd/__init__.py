"""egdst_amd: MI355X-native DC-EGM solver + simulator behind the egdstmodel surface."""
import os as _os

# A batched handle runs its draw groups on 4 to 8 HIP streams (include/egdst.h: egdst_set_groups).  The HIP runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, read once when the runtime starts); with the default the
# group streams share queues with the caller's stream and the null stream and overlap poorly (C2, 4096 draws per
# solve on MI355X: 890 ms with 4 queues, 708 ms with 8, and 8 groups on 12 queues beat 4 groups on 8 by another 7 % --
# DESIGN.md section 5).  Ask for 12 unless the user chose.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '12')

from .model import egdstmodel, EgdstError  # noqa: F401
from .quadrature import quadpoints  # noqa: F401

__all__ = ['egdstmodel', 'EgdstError', 'quadpoints']
