"""`networks` as GenNet/predict.py:13 and GenNet/train.py:7 import it (`from networks import AEViT as AE`;
GenNet/networks/__init__.py:1-3 re-exports its sub-modules' names).  With `ppnet_amd/dropin` first on sys.path the
reference's predict.py builds the HIP-backed AE-ViT: same constructor (`AE(img_channels, out_channels, img_resolution,
dim)`), same 87 state-dict keys, `load_state_dict(torch.load(p)['model'])` unchanged.  `token2feature` / `feature2token`
are the two helpers of networks/base.py:43-52 that scripts import by name.  `AESwin` (the commented-out alternative,
predict.py:12) is out of scope: asking for it raises with that message instead of an ImportError further down."""
from ppnet_amd.gennet import AE, AEViT, normalize_heatmap_u8  # noqa: F401


def token2feature(x, H, W):
    """[B, N, C] -> [B, C, H, W] (networks/base.py:43-46)."""
    B, N, C = x.shape
    return x.permute(0, 2, 1).reshape(B, C, H, W)


def feature2token(x):
    """[B, C, H, W] -> [B, H*W, C] (networks/base.py:49-52)."""
    B, C, H, W = x.shape
    return x.view(B, C, -1).transpose(1, 2)


def __getattr__(name):
    if name == "AESwin":
        raise NotImplementedError("networks.AESwin: the shifted-window variant is not the selected model "
                                  "(GenNet/predict.py:12 keeps it commented out) and is out of this build's scope")
    raise AttributeError(name)


__all__ = ["AE", "AEViT", "token2feature", "feature2token", "normalize_heatmap_u8"]
