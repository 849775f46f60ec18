"""Layer tables of the Voxception-ResNet codec.

Restates the layer lists of the reference's models/model_voxception.py
(_VoxceptionResNet 11-68, AnalysisTransform 71-144, SynthesisTransform 147-214,
HyperEncoder 217-252, HyperDecoder 255-308) as data: one `Layer` per Keras
layer, keyed by the Keras *attribute* path, which is also the
tf.train.Checkpoint key (e.g. ``vrn1_1/conv1_1/kernel``).

kind: "conv" (Conv3D, kernel [k,k,k,Cin,Cout]) or "tconv" (Conv3DTranspose,
kernel [k,k,k,Cout,Cin]); all padding='same', channels-last, float32.
"""
from collections import namedtuple

Layer = namedtuple("Layer", "name kind cin cout k stride bias relu")


def _vrn(name, c):
    q, h = c // 4, c // 2
    return [
        Layer(name + "/conv1_1", "conv", c, q, 3, 1, True, True),
        Layer(name + "/conv1_2", "conv", q, h, 3, 1, True, True),
        Layer(name + "/conv2_1", "conv", c, q, 1, 1, True, True),
        Layer(name + "/conv2_2", "conv", q, q, 3, 1, True, True),
        Layer(name + "/conv2_3", "conv", q, h, 1, 1, True, True),
    ]


def analysis_layers():
    L = [Layer("conv_in", "conv", 1, 16, 3, 1, True, True)]
    for i in (1, 2, 3):
        L += _vrn("vrn1_%d" % i, 16)
    L.append(Layer("down_1", "conv", 16, 32, 3, 2, False, True))
    for i in (1, 2, 3):
        L += _vrn("vrn2_%d" % i, 32)
    L.append(Layer("down_2", "conv", 32, 64, 3, 2, False, True))
    for i in (1, 2, 3):
        L += _vrn("vrn3_%d" % i, 64)
    L.append(Layer("conv_out", "conv", 64, 16, 3, 1, True, False))
    return L


def synthesis_layers():
    L = [Layer("deconv_in", "conv", 16, 64, 3, 1, True, True)]
    for i in (1, 2, 3):
        L += _vrn("vrn1_%d" % i, 64)
    L.append(Layer("up_1", "tconv", 64, 32, 3, 2, True, True))
    for i in (1, 2, 3):
        L += _vrn("vrn2_%d" % i, 32)
    L.append(Layer("up_2", "tconv", 32, 16, 3, 2, True, True))
    for i in (1, 2, 3):
        L += _vrn("vrn3_%d" % i, 16)
    L.append(Layer("deconv_out", "conv", 16, 1, 3, 1, True, False))
    return L


def hyper_encoder_layers():
    return [
        Layer("conv1", "conv", 16, 16, 3, 1, True, True),
        Layer("conv2", "conv", 16, 16, 3, 2, True, True),
        Layer("conv3", "conv", 16, 8, 3, 1, True, False),
    ]


def hyper_decoder_layers():
    return [
        Layer("conv1", "conv", 8, 16, 3, 1, True, True),
        Layer("conv2", "tconv", 16, 16, 3, 2, True, True),
        Layer("conv3", "conv", 16, 32, 3, 1, True, True),
        Layer("conv4_1", "conv", 32, 16, 3, 1, True, False),
        Layer("conv4_2", "conv", 32, 16, 3, 1, True, False),
    ]


# models/model_simple.py:12-49 (analysis: 9^3 s2, 5^3 s2, 5^3 s2, 32 channels; the last layer linear without bias)
# and 52-95 (synthesis: 5^3, 5^3, 9^3 transposed s2; the last one linear).  64^3 x 1 -> 8^3 x 32 -> 64^3 x 1.
def simple_analysis_layers():
    return [
        Layer("conv_1", "conv", 1, 32, 9, 2, True, True),
        Layer("conv_2", "conv", 32, 32, 5, 2, True, True),
        Layer("conv_3", "conv", 32, 32, 5, 2, False, False),
    ]


def simple_synthesis_layers():
    return [
        Layer("deconv_1", "tconv", 32, 32, 5, 2, True, True),
        Layer("deconv_2", "tconv", 32, 32, 5, 2, True, True),
        Layer("deconv_3", "tconv", 32, 1, 9, 2, True, False),
    ]


SIMPLE_NETS = {
    "analysis_transform": simple_analysis_layers,
    "synthesis_transform": simple_synthesis_layers,
}

NETS = {
    "analysis_transform": analysis_layers,
    "synthesis_transform": synthesis_layers,
    "hyper_encoder": hyper_encoder_layers,
    "hyper_decoder": hyper_decoder_layers,
}


def kernel_shape(layer):
    if layer.kind == "conv":
        return (layer.k, layer.k, layer.k, layer.cin, layer.cout)
    return (layer.k, layer.k, layer.k, layer.cout, layer.cin)


def macs_per_cube(net, cube_size=64):
    """Multiply-accumulates of one forward pass on one cube (bias/activation excluded).
    Every layer that is not a resampler has stride 1, so a running spatial size suffices."""
    n = {"analysis_transform": cube_size, "synthesis_transform": cube_size // 4,
         "hyper_encoder": cube_size // 4, "hyper_decoder": cube_size // 8}[net]
    total = 0
    for l in NETS[net]():
        if l.kind == "conv":
            n = n // l.stride
            total += (n ** 3) * (l.k ** 3) * l.cin * l.cout
        else:
            total += (n ** 3) * (l.k ** 3) * l.cin * l.cout
            n = n * 2
    return total
