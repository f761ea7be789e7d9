"""Flag surface of the reference's ``main.py`` (main.py:14-43: same names, types and defaults -- SURVEY.md 8(f)3)
driving ``sggan`` on synthetic batches.  The reference's quirks are kept as data, not behaviour: --lr, --L1_lambda
and friends are parsed but the live (reference-mode) step hard-codes lr=1e-3 and LAMBDA=100 (model.py:151,205);
they take effect in --cycle mode, which uses the criteria those flags were written for."""
from __future__ import annotations

import argparse


def build_parser():
    p = argparse.ArgumentParser(description="")
    a = p.add_argument
    a("--dataset_dir", dest="dataset_dir", default="city")
    a("--epoch", dest="epoch", type=int, default=100)
    a("--epoch_step", dest="epoch_step", type=int, default=100)
    a("--batch_size", dest="batch_size", type=int, default=1)
    a("--train_size", dest="train_size", type=int, default=int(1e8))
    a("--img_height", dest="image_height", type=int, default=64)
    a("--img_width", dest="image_width", type=int, default=64)
    a("--ratio_gan2seg", dest="ratio_gan2seg", type=int, default=10)
    a("--use_augmentation", dest="use_augmentation", type=bool, default=True)
    a("--ngf", dest="ngf", type=int, default=64)
    a("--ndf", dest="ndf", type=int, default=64)
    a("--input_nc", dest="input_nc", type=int, default=3)
    a("--output_nc", dest="output_nc", type=int, default=3)
    a("--lr", dest="lr", type=float, default=0.0002)
    a("--beta1", dest="beta1", type=float, default=0.5)
    a("--which_direction", dest="which_direction", default="AtoB")
    a("--phase", dest="phase", default="train")
    a("--save_freq", dest="save_freq", type=int, default=1000)
    a("--print_freq", dest="print_freq", type=int, default=5)
    a("--continue_train", dest="continue_train", type=bool, default=False)
    a("--checkpoint_dir", dest="checkpoint_dir", default="./checkpoint")
    a("--sample_dir", dest="sample_dir", default="./sample")
    a("--test_dir", dest="test_dir", default="./test")
    a("--L1_lambda", dest="L1_lambda", type=float, default=10.0)
    a("--Lg_lambda", dest="Lg_lambda", type=float, default=5.0)
    a("--use_resnet", dest="use_resnet", type=bool, default=False)
    a("--use_lsgan", dest="use_lsgan", type=bool, default=True)
    a("--use_pix2pix", dest="use_pix2pix", type=bool, default=False)
    a("--max_size", dest="max_size", type=int, default=50)
    a("--segment_class", dest="segment_class", type=int, default=34)
    # build-specific knobs (not in the reference)
    a("--cycle", dest="cycle", action="store_true", help="2G+2D cycle-mode step (north_star unit)")
    a("--dtype", dest="dtype", default="bf16")
    a("--steps_per_epoch", dest="steps_per_epoch", type=int, default=4)
    a("--use_pool", dest="use_pool", action="store_true",
      help="train the discriminators on utils.ImagePool's history of fakes (--max_size entries; upstream SG-GAN behaviour, cycle mode)")
    a("--graph", dest="graph", action="store_true", help="replay the step from captured HIP graphs")
    a("--checkpoint_blocks", dest="checkpoint_blocks", action="store_true",
      help="activation checkpointing: re-run each residual block in backward instead of keeping its activations")
    a("--log_dir", dest="log_dir", default="./logs", help="scalar summaries (the reference writes tfevents under logs/<timestamp>/train)")
    return p


def synthetic_batches(model, args):
    import torch
    from .segment_class import one_hot_mask

    def gen(epoch):
        g = torch.Generator().manual_seed(19 + epoch)
        H, W, N = args.image_height, args.image_width, args.batch_size
        mh, mw = model.discriminator.out_hw(H, W)
        if (mh, mw) == (1, 1):
            mh, mw = round(H / 34), round(W / 34)
        for _ in range(args.steps_per_epoch):
            def dom():
                idx = torch.randint(0, args.segment_class, (N, max(H // 32, 1), max(W // 32, 1)), generator=g)
                idx = idx.repeat_interleave(H // idx.shape[1], 1).repeat_interleave(W // idx.shape[2], 2).to(torch.uint8)
                pal = torch.rand((args.segment_class, 3), generator=torch.Generator().manual_seed(7))
                return torch.rand((N, H, W, 3), generator=g), pal[idx.long()], one_hot_mask(idx, mh, mw, args.segment_class)
            rA, sA, mA = dom()
            b = {"real_A": rA, "seg_A": sA, "mask_A": mA}
            if args.cycle:
                rB, sB, mB = dom()
                b.update({"real_B": rB, "seg_B": sB, "mask_B": mB})
            yield b
    return gen


def synthetic_test_samples(args, count=2):
    """(name, sample_image, seg_image) triples in the reference loader's ranges (utils.load_test_data: floats in [0,1])."""
    import torch

    def gen(epoch=0):
        g = torch.Generator().manual_seed(1000 + epoch)
        H, W = args.image_height, args.image_width
        for i in range(count):
            yield "synthetic_%03d.png" % i, torch.rand((H, W, 3), generator=g).numpy(), torch.rand((H, W, 3), generator=g).numpy()
    return gen


def main(argv=None):
    """main.py:45-60: ``--phase train`` runs the epoch loop (with the epoch-end test pass and scalar summaries of
    model.py:263-268), ``--phase test`` the test pass of model.py:535-567 -- here on synthetic data (dataset files and their
    decoding are host-side I/O outside the hot path)."""
    args = build_parser().parse_args(argv)
    args.use_resnet = True            # the only generator on the hot path
    from .model import sggan
    from .utils import SummarySink
    model = sggan(args)
    if args.phase == "test":          # main.py:58-60
        return model.test(args, synthetic_test_samples(args)())
    import os
    sink = SummarySink(os.path.join(getattr(args, "log_dir", "./logs"), "train", "scalars.jsonl"))
    return model.train(args, synthetic_batches(model, args), test_samples=synthetic_test_samples(args), sink=sink)


if __name__ == "__main__":
    main()
