"""Mirror of model/segformer_head.py: SegFormerHead (:47-77) as a weight holder over the HIP decoder handle.

The head and the decoder transformer run inside ONE library call (excel_decoder_forward), so this class only carries
the state_dict; ExCEL_model wires both into an ops.DecoderHandle."""


class SegFormerHead:
    def __init__(self, in_channels=128, embedding_dim=256, num_classes=20, index=11, **kwargs):
        self.in_channels, self.embedding_dim, self.num_classes, self.indexes = in_channels, embedding_dim, num_classes, index
        self._sd = None

    def load_state_dict(self, sd, strict=True):
        need = [f"linears_modulelist.{l}.{k}" for l in range(self.indexes) for k in ("proj.weight", "proj.bias", "proj_2.weight", "proj_2.bias")]
        need += ["linear_fuse.weight", "linear_fuse.bias"]
        missing = [k for k in need if k not in sd]
        if missing and strict:
            raise KeyError(f"SegFormerHead.load_state_dict: missing {missing[:3]}...")
        self._sd = dict(sd)
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def eval(self):
        return self
