"""open_clip.create_model_and_transforms for the golden generator: a CLIP-like object whose `.visual` is the oracle's
restatement of open_clip's VisionTransformer (tokens returned before ln_post, `proj` / `output_tokens` attributes)"""
import e4t_oracle as _orc

TEST_ARCHS = {"ViT-golden-test": dict(image_size=28, patch_size=14, width=16, layers=2, heads=2, mlp_ratio=2.0)}


class _Clip:
    def __init__(self, vit_cfg):
        self.visual = _orc.VisionTransformer(**vit_cfg)
        self.transformer = None


def create_model_and_transforms(arch, device=None, pretrained=None):
    return _Clip(TEST_ARCHS[arch]), None, None
