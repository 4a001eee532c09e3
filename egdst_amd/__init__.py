"""egdst_amd: MI355X-native DC-EGM solver + simulator behind the egdstmodel surface."""
import os as _os

# A batched handle runs its draw groups on 4 to 16 HIP streams (include/egdst.h: egdst_set_groups).  The HIP runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, read once when the runtime starts); with the default the
# group streams share queues with the caller's stream and the null stream and overlap poorly (C2, 4096 draws per
# solve on MI355X, 4 groups: 890 ms with 4 queues, 708 ms with 8; DESIGN.md section 5 has the sweep up to 16 groups on
# 24 queues).  Ask for 24 unless the user chose.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')

from .model import egdstmodel, EgdstError  # noqa: F401
from .quadrature import quadpoints  # noqa: F401

__all__ = ['egdstmodel', 'EgdstError', 'quadpoints']
