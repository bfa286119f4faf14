"""TEST HELPER: the pixel conv stacks through torch autograd (MIOpen on the device) behind the interface of
big_dreamer_amd.conv_stack.ConvStacks -- the comparator the product engine carried as a second backend in rounds 1-2
(BD_CONV=miopen).  Inject with ``eng.conv = TorchConvStacks(eng)`` right after building a pixel engine: the engine's
schedule, losses, weight-gradient batches and optimiser are unchanged, only encoder / decoder forward and backward come from
``torch.nn.functional`` on the engine's own parameters (logical NCHW views of the permuted storage).

CnnImageEncoder src/models.py:527-564; ObservationModel src/models.py:319-362."""
import torch
import torch.nn.functional as Fnn


class TorchConvStacks:
    def __init__(self, eng):
        self.e = eng
        self.lin_tail = eng.d.E != 1024

    def pack(self):            # nothing packed: torch reads the parameter views directly
        pass

    def _leaves(self, mod):
        e = self.e
        g = e.groups[e._mod_group[mod]]
        return {n: g.p[(m, n)].detach().contiguous().requires_grad_(True) for (m, n, _) in g.specs if m == mod}

    def encode(self, obs4d, tag=""):
        grad = tag == ""
        with torch.set_grad_enabled(grad):
            w = self._leaves("encoder")
            x = obs4d
            for i in range(4):
                x = Fnn.elu(Fnn.conv2d(x, w[f"model.{2 * i}.weight"], w[f"model.{2 * i}.bias"], stride=2))
            x = x.flatten(1)
            if "model.9.weight" in w:
                x = Fnn.linear(x, w["model.9.weight"], w["model.9.bias"])
        if grad:
            self._enc_graph = (x, w)
            self.acts_enc = [obs4d.permute(0, 2, 3, 1).contiguous()]      # NHWC targets, as ConvStacks keeps them
        return x.detach().contiguous()

    def decode(self, feat, tag=""):
        grad = tag == ""
        with torch.set_grad_enabled(grad):
            w = self._leaves("observation_model")
            f = feat.detach().requires_grad_(grad)
            x = Fnn.linear(f, w["decoder.0.weight"], w["decoder.0.bias"]).view(feat.shape[0], -1, 1, 1)
            for idx in (2, 4, 6, 8):
                x = Fnn.conv_transpose2d(x, w[f"decoder.{idx}.weight"], w[f"decoder.{idx}.bias"], stride=2)
                if idx != 8:
                    x = Fnn.elu(x)
        if grad:
            self._dec_graph = (x, w, f)
        return x.detach().permute(0, 2, 3, 1).contiguous()                  # NHWC prediction

    def backward_decoder(self, g_pred, feat, dfeat, wb):
        e = self.e
        pred, w, f = self._dec_graph
        names = list(w)
        grads = torch.autograd.grad(pred, [w[n] for n in names] + [f], g_pred.permute(0, 3, 1, 2).contiguous())
        for n, g in zip(names, grads[:-1]):
            e.G("observation_model", n).copy_(g)
        dfeat.add_(grads[-1])
        self._dec_graph = None

    def backward_encoder(self, d_emb, wb):
        e = self.e
        x, w = self._enc_graph
        names = list(w)
        for n, g in zip(names, torch.autograd.grad(x, [w[n] for n in names], d_emb)):
            e.G("encoder", n).copy_(g)
        self._enc_graph = None
