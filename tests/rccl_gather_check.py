"""Body of tests/test_gpu_parity.py::test_result_gather_over_rccl_is_host_issued_and_exact (own process: see there)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd.core import distributed as D  # noqa: E402
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
        cams, seq = S.load_cameras("panoptic")
        rt = S.resize_transform(cfg).to("cuda:0")
        model = FV.get(cfg).to("cuda:0")
        model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=13))
        B = 2
        heats = [S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=70 + i).to("cuda:0") for i in range(4)]
        meta = {"seq": [seq] * B}
        gat = D.ResultGatherer(1, device="cuda:0", always=True)
        assert gat.issue == "host"
        pipe = FV.PipelinedForward(model, depth=4)
        local, gathered, issued = [], [], []
        with torch.no_grad():
            for i in range(16):
                (fused, _, _, _, _), ev = pipe.submit(meta=meta, input_heatmaps=heats[i % 4], cameras=cams, resize_transform=rt)
                local.append(fused)
                gathered.append(gat.gather(fused, ev))
                issued.append(i + 1 - len(gat._pending))
            gat.synchronize()
            torch.cuda.synchronize()
        assert not gat._pending
        # the last eight results are still in the ring (8 buffers): equal to the batches' own outputs
        for i in range(8, 16):
            assert gathered[i].shape == local[i].shape and torch.equal(gathered[i], local[i]), i
        assert gathered[15].data_ptr() != local[15].data_ptr()
        # streaming: gathers were issued inside the submit loop, never more than `depth` batches behind
        assert issued[-1] >= 16 - 4 and all(b - a <= 4 for a, b in zip(issued, range(1, 17))), issued
        print("RCCL GATHER OK", issued)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
