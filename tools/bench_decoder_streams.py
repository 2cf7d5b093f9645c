#!/usr/bin/env python3
"""BASELINE configs[3] alone (128 streams, bf16, hipGraph decode steps) -- for rocprofv3 runs."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from etude_amd import synth  # noqa: E402
from etude_amd.decoder import EtudeDecoderConfig  # noqa: E402

if __name__ == "__main__":
    ctx0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    print(json.dumps(bench.decoder_stream_bench(EtudeDecoderConfig(**synth.decoder_dims()), dev, ctx0=ctx0)))
