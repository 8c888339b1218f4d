"""scp_amd.data -- the training input pipeline: on-disk Wild6D layout -> device batches."""
from .wild6d import GpuCollator, Wild6DDataset, Wild6DTestDataset, data_loader, test_loader  # noqa: F401
