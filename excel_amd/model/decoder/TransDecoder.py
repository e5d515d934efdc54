"""Mirror of model/decoder/TransDecoder.py: DecoderTransformer (:105-124) as a weight holder (see segformer_head.py)."""


class DecoderTransformer:
    def __init__(self, width, layers, heads, output_dim):
        self.width, self.layers, self.heads, self.output_dim = width, layers, heads, output_dim
        self._sd = None

    def load_state_dict(self, sd, strict=True):
        need = [f"transformer.resblocks.{l}.{k}" for l in range(self.layers)
                for k in ("ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                          "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                          "mlp.c_proj.weight", "mlp.c_proj.bias")] + ["linear_pred.weight", "linear_pred.bias"]
        missing = [k for k in need if k not in sd]
        if missing and strict:
            raise KeyError(f"DecoderTransformer.load_state_dict: missing {missing[:3]}...")
        self._sd = dict(sd)
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def eval(self):
        return self
