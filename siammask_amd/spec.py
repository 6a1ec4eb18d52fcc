"""Declarative description of the three SiamMask inference networks.

This is the *state-dict contract* of SURVEY.md Appendix B: every parameter / buffer name
and shape the reference's ``Custom`` modules expose, so that a checkpoint written for the
reference loads through the reference's unchanged ``utils/load_helper.py:30-54`` into the
drop-in ``siammask_amd.custom.Custom`` with zero missing keys.

Nothing here is copied from the reference; the table is derived from the architecture
described by
  * experiments/siammask_sharp/resnet.py:59-103,151-227   (Bottleneck / ResNet layout)
  * experiments/siammask_sharp/custom.py:12-25,69-129     (ResDownS / UP / MaskCorr / Refine)
  * models/rpn.py:41-61                                   (DepthCorr)
and verified against the instantiated reference modules by tests/test_spec.py
(356 / 324 / 304 entries for sharp / base / rpn).
"""
from collections import OrderedDict, namedtuple

VARIANTS = ("rpn", "base", "sharp")

# (layer index, planes, blocks, stride of block 0, dilation of blocks 1..)
RESNET_STAGES = ((1, 64, 3, 1, 1), (2, 128, 4, 2, 1), (3, 256, 6, 1, 2))

ConvSpec = namedtuple(
    "ConvSpec", "name cin cout k stride pad dil bn bias relu")
# name : state-dict prefix of the conv ("<name>.weight")
# bn   : state-dict prefix of the BatchNorm that follows it, or None
# bias : True if the conv has its own bias ("<name>.bias")


def backbone_convs():
    """Every conv of the modified ResNet-50 + adjust layer, in execution order.

    Geometry per experiments/siammask_sharp/resnet.py:
      conv1 7x7 s2 p0 (:154), Bottleneck conv2 padding = 2 - stride, or = dilation when
      dilation > 1 (:66-72), shortcut convs per _make_layer (:184-206).
    """
    f = "features.features."
    out = [ConvSpec(f + "conv1", 3, 64, 7, 2, 0, 1, f + "bn1", False, True)]
    inplanes = 64
    for idx, planes, blocks, stride, dilation in RESNET_STAGES:
        for b in range(blocks):
            p = "%slayer%d.%d." % (f, idx, b)
            if b == 0:
                s = stride
                # block 0 runs with dilation dd (resnet.py:187-210)
                if stride == 1 and dilation == 1:
                    dd = 1
                    ds = ConvSpec(p + "downsample.0", inplanes, planes * 4, 1, 1, 0, 1,
                                  p + "downsample.1", False, False)
                else:
                    if dilation > 1:
                        dd = dilation // 2
                        ds_pad = dd
                    else:
                        dd = 1
                        ds_pad = 0
                    ds = ConvSpec(p + "downsample.0", inplanes, planes * 4, 3, stride, ds_pad, dd,
                                  p + "downsample.1", False, False)
                d = dd
            else:
                s, d, ds = 1, dilation, None
            pad2 = d if d > 1 else 2 - s
            cin = inplanes if b == 0 else planes * 4
            out.append(ConvSpec(p + "conv1", cin, planes, 1, 1, 0, 1, p + "bn1", False, True))
            out.append(ConvSpec(p + "conv2", planes, planes, 3, s, pad2, d, p + "bn2", False, True))
            # conv3's ReLU comes after the residual add
            out.append(ConvSpec(p + "conv3", planes, planes * 4, 1, 1, 0, 1, p + "bn3", False, False))
            if ds is not None:
                out.append(ds)
        inplanes = planes * 4
    out.append(ConvSpec("features.downsample.downsample.0", 1024, 256, 1, 1, 0, 1,
                        "features.downsample.downsample.1", False, False))
    return out


def depthcorr_convs(prefix, out_channels):
    """models/rpn.py:41-61 — conv_kernel / conv_search / head of one DepthCorr."""
    return [
        ConvSpec(prefix + "conv_kernel.0", 256, 256, 3, 1, 0, 1, prefix + "conv_kernel.1", False, True),
        ConvSpec(prefix + "conv_search.0", 256, 256, 3, 1, 0, 1, prefix + "conv_search.1", False, True),
        ConvSpec(prefix + "head.0", 256, 256, 1, 1, 0, 1, prefix + "head.1", False, True),
        ConvSpec(prefix + "head.3", 256, out_channels, 1, 1, 0, 1, None, True, False),
    ]


def refine_convs():
    """experiments/siammask_sharp/custom.py:102-124 (all 3x3 p1, with bias)."""
    r = "refine_model."
    out = []
    for name, chans in (("v0", (64, 16, 4)), ("v1", (256, 64, 16)), ("v2", (512, 128, 32)),
                        ("h2", (32, 32, 32)), ("h1", (16, 16, 16)), ("h0", (4, 4, 4))):
        out.append(ConvSpec("%s%s.0" % (r, name), chans[0], chans[1], 3, 1, 1, 1, None, True, True))
        out.append(ConvSpec("%s%s.2" % (r, name), chans[1], chans[2], 3, 1, 1, 1, None, True, True))
    # deconv is a ConvTranspose2d(256, 32, 15, 15): weight (256, 32, 15, 15), listed separately
    for name, cin, cout in (("post0", 32, 16), ("post1", 16, 4), ("post2", 4, 1)):
        out.append(ConvSpec(r + name, cin, cout, 3, 1, 1, 1, None, True, False))
    return out


def all_convs(variant):
    assert variant in VARIANTS
    convs = backbone_convs()
    convs += depthcorr_convs("rpn_model.cls.", 10)
    convs += depthcorr_convs("rpn_model.loc.", 20)
    if variant in ("base", "sharp"):
        convs += depthcorr_convs("mask_model.mask.", 63 * 63)
    if variant == "sharp":
        convs += refine_convs()
    return convs


def state_dict_spec(variant):
    """OrderedDict name -> (shape tuple, kind) with kind in
    {'conv_w','bias','bn_w','bn_b','bn_mean','bn_var','bn_nbt','deconv_w'}."""
    spec = OrderedDict()

    def add_conv(c):
        spec[c.name + ".weight"] = ((c.cout, c.cin, c.k, c.k), "conv_w")
        if c.bias:
            spec[c.name + ".bias"] = ((c.cout,), "bias")
        if c.bn:
            spec[c.bn + ".weight"] = ((c.cout,), "bn_w")
            spec[c.bn + ".bias"] = ((c.cout,), "bn_b")
            spec[c.bn + ".running_mean"] = ((c.cout,), "bn_mean")
            spec[c.bn + ".running_var"] = ((c.cout,), "bn_var")
            spec[c.bn + ".num_batches_tracked"] = ((), "bn_nbt")

    for c in all_convs(variant):
        if c.name == "refine_model.post0":
            spec["refine_model.deconv.weight"] = ((256, 32, 15, 15), "deconv_w")
            spec["refine_model.deconv.bias"] = ((32,), "bias")
        add_conv(c)
    return spec


def module_tree(variant):
    """Nested dict mirroring the reference module hierarchy (leaf = list of (attr, shape, kind))."""
    tree = OrderedDict()
    for name, (shape, kind) in state_dict_spec(variant).items():
        parts = name.split(".")
        node = tree
        for p in parts[:-1]:
            node = node.setdefault(p, OrderedDict())
        node[parts[-1]] = (shape, kind)
    return tree


# ---- sizes the outputs must line up with -------------------------
SEARCH_SIZE = 255
TEMPLATE_SIZE = 127
SCORE_SIZE = 25          # (255-127)//8 + 1 + 8, utils/tracker_config.py:23
ANCHOR_NUM = 5
MASK_OUT = 63            # base: 63x63 logits per position
REFINE_OUT = 127         # sharp: 127x127 refined mask

# algorithmic FLOPs per frame (2*MAC, BN folded, conv_kernel cached)
GFLOP_PER_FRAME = {"rpn": 30.870, "base": 33.222, "sharp": 33.915}
GFLOP_TEMPLATE = 6.824
