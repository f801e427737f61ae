"""Dataset front-end: DTU and BlendedMVS item dictionaries without OpenCV.  `get_loader` mirrors
/root/reference/datasets/__init__.py:16-38 (the four datasets it knows: generalisation and per-scene fine-tune, DTU and BlendedMVS)."""
import torch.distributed as dist
from torch.utils.data import DataLoader, DistributedSampler, RandomSampler, SequentialSampler

from .bmvs import BMVSDataset  # noqa: F401
from .bmvs_finetune import BMVSDatasetFinetune  # noqa: F401
from .dtu import DTUDataset  # noqa: F401
from .dtu_finetune import DTUDatasetFinetune  # noqa: F401


def collect_fn(data):
    return data[0]


def get_loader(conf, mode, distributed):
    name = conf.get_string("dataset_name")
    if name == "DTUDataset":
        dataset = DTUDataset(conf, mode)
    elif name == "DTUDatasetFinetune":
        dataset = DTUDatasetFinetune(conf, mode)
    elif name == "BMVSDataset":
        dataset = BMVSDataset(conf, mode)
    elif name == "BMVSDatasetFinetune":
        dataset = BMVSDatasetFinetune(conf, mode)
    else:
        raise NotImplementedError(name)
    if mode == "finetune":
        return dataset
    if distributed:
        sampler = DistributedSampler(dataset, num_replicas=dist.get_world_size(), rank=dist.get_rank())
    else:
        sampler = RandomSampler(dataset) if mode == "train" else SequentialSampler(dataset)
    loader = DataLoader(dataset, 1, sampler=sampler, num_workers=8, drop_last=(mode == "train"), pin_memory=False, collate_fn=collect_fn)
    return loader, sampler, dataset
