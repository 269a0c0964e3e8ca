#!/usr/bin/env python3
"""A/B baselines that are NOT product paths: bench.py with the ResNet bottlenecks on the library kernels (ATen -> MIOpen / rocBLAS
convolutions + the fused bn_act launch) and / or torch's own fused SGD, patched in from here.

    python tools/bench_library_paths.py --backbone aten --optim aten [bench.py arguments]

(Rounds 1-3 carried these as CIM_BACKBONE_1X1 / CIM_BACKBONE_3X3 / CIM_OPTIM switches inside the product.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    argv = sys.argv[1:]
    take = lambda flag: (argv.pop(argv.index(flag) + 1), argv.pop(argv.index(flag)))[0] if flag in argv else "hip"
    backbone, optim = take("--backbone"), take("--optim")
    os.environ["CIM_ALLOW_FALLBACK"] = "*"                 # library branches are the point here (an error by default)
    import torch
    import bench
    if backbone == "aten":
        from cim_amd.modeling import resnet50
        from cim_amd.ops import bn_act

        def forward(self, x):
            identity = x
            out = bn_act(self.conv1(x), self.bn1)
            out = bn_act(self.conv2(out), self.bn2)
            if self.downsample is not None:
                identity = bn_act(self.downsample[0](x), self.downsample[1], relu=False)
            return bn_act(self.conv3(out), self.bn3, residual=identity)
        resnet50.Bottleneck.forward = forward
    if optim == "aten":
        def make_optimizer(model, torch=torch):
            bias, nonbias = [], []
            for name, p in model.named_parameters():
                if p.requires_grad:
                    (bias if "bias" in name else nonbias).append(p)
            lr, wd = 0.0005, 0.0005
            return torch.optim.SGD([dict(params=nonbias, lr=lr, weight_decay=wd), dict(params=bias, lr=2 * lr, weight_decay=0.0)],
                                   lr=lr, momentum=0.9, fused=True)
        bench.make_optimizer = make_optimizer
    sys.argv = [os.path.join(os.path.dirname(bench.__file__), "bench.py")] + argv
    bench.main()


if __name__ == "__main__":
    main()
