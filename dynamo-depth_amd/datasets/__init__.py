from .kitti_dataset import KITTIDataset  # noqa: F401
from .synthetic import SyntheticTriplets, WaymoDataset, nuScenesDataset  # noqa: F401
