"""jittor.nn for the shim (see __init__.py): Module / ModuleList / Linear / activations / init / the Optimizer base class."""
import math
import torch


class Module:
    """jittor: a plain Python class - subclasses need not call super().__init__() (HuberLoss, FrequencyEncoder do not), `execute` is the forward"""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self.execute(*a, **k)

    def _walk(self, prefix=""):
        for k, v in vars(self).items():
            if isinstance(v, torch.Tensor):
                yield prefix + k, v
            elif isinstance(v, Module):
                yield from v._walk(prefix + k + ".")
            elif isinstance(v, (list, tuple)):
                for i, m in enumerate(v):
                    if isinstance(m, Module):
                        yield from m._walk(prefix + k + "." + str(i) + ".")

    def named_parameters(self):
        return list(self._walk())

    def parameters(self):
        return [v for _, v in self._walk()]

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self._walk()}

    def train(self):
        pass

    def eval(self):
        pass


class ModuleList(Module, list):
    def __init__(self, mods=()):
        list.__init__(self, mods)

    def _walk(self, prefix=""):
        for i, m in enumerate(self):
            yield from m._walk(prefix + str(i) + ".")


class Sequential(Module):
    """jittor: nn.Sequential - indexable, modules applied in order"""
    def __init__(self, *mods):
        self.layers = list(mods)

    def __getitem__(self, i):
        return self.layers[i]

    def __len__(self):
        return len(self.layers)

    def execute(self, x):
        for m in self.layers:
            x = m(x)
        return x

    def float16(self):
        return self


class Linear(Module):
    """jittor: nn.Linear - weight [out, in] ~ init.invariant_uniform = U(+-sqrt(3 / fan_in)), bias ~ U(+-1 / sqrt(fan_in))"""
    def __init__(self, in_features, out_features, bias=True):
        bw = math.sqrt(3.0 / in_features)
        self.weight = (torch.rand(out_features, in_features) * 2 - 1) * bw
        self.bias = (torch.rand(out_features) * 2 - 1) / math.sqrt(in_features) if bias else None

    def execute(self, x):
        y = x @ self.weight.t()
        return y + self.bias if self.bias is not None else y


def relu(x):
    return torch.relu(x)


def softplus(x, beta=1.0, threshold=20.0):
    """jittor: nn.softplus = 1/beta * log(1 + exp(beta x)), linear where beta x > threshold"""
    return torch.nn.functional.softplus(x, beta=beta, threshold=threshold)


class ReLU(Module):
    def execute(self, x):
        return torch.relu(x)


class Softplus(Module):
    def __init__(self, beta=1, threshold=20):
        self.beta, self.threshold = beta, threshold

    def execute(self, x):
        return softplus(x, self.beta, self.threshold)


class init:
    @staticmethod
    def gauss_(var, mean=0.0, std=1.0):
        return torch.randn(var.shape) * std + mean

    @staticmethod
    def constant_(var, value=0.0):
        return torch.full(var.shape, float(value))


class Optimizer:
    """jittor: nn.Optimizer(params, lr) - param_groups is a list of dicts with "params"; a bare list of Vars becomes one group.  step(loss) differentiates the SUM of
    `loss` w.r.t. the parameters (jt.grad of a non-scalar) before the update."""
    def __init__(self, params, lr, param_sync_iter=10000):
        self.lr = lr
        params = list(params)
        if params and not isinstance(params[0], dict):
            params = [{"params": params}]
        self.param_groups = params
        for pg in self.param_groups:
            pg["params"] = [p.requires_grad_(True) for p in pg["params"]]
        self.n_step = 0

    def zero_grad(self):
        pass

    def pre_step(self, loss, retain_graph=False):
        if loss is not None:
            params = [p for pg in self.param_groups for p in pg["params"]]
            grads = torch.autograd.grad(loss.sum(), params, allow_unused=True, retain_graph=retain_graph)
            it = iter(grads)
            for pg in self.param_groups:
                pg["grads"] = [g if g is not None else torch.zeros_like(p) for p, g in zip(pg["params"], (next(it) for _ in pg["params"]))]
        self.n_step += 1

    def backward(self, loss, retain_graph=False):
        self.pre_step(loss, retain_graph)
        self.n_step -= 1

    def state_dict(self):
        return {"defaults": getattr(self, "defaults", {})}

    def load_state_dict(self, sd):
        pass


class Adam(Optimizer):
    """jittor: nn.Adam (restated from Jittor's optim code, NOT pinned - it lives inside Jittor): "values" is the second moment, "m" the first;
    step_size = lr * sqrt(1 - b1^n) / (1 - b0^n);  p -= m * step_size / (sqrt(values) + eps)"""
    def __init__(self, params, lr, eps=1e-8, betas=(0.9, 0.999), weight_decay=0):
        super().__init__(params, lr)
        self.eps, self.betas, self.weight_decay = eps, betas, weight_decay
        for pg in self.param_groups:
            pg["values"] = [torch.zeros_like(p) for p in pg["params"]]
            pg["m"] = [torch.zeros_like(p) for p in pg["params"]]

    def step(self, loss=None, retain_graph=False):
        self.pre_step(loss, retain_graph)
        n = float(self.n_step)
        b0, b1 = self.betas
        with torch.no_grad():
            for pg in self.param_groups:
                lr = pg.get("lr", self.lr)
                for p, g, v, m in zip(pg["params"], pg["grads"], pg["values"], pg["m"]):
                    g = p * self.weight_decay + g
                    m.copy_(b0 * m + (1 - b0) * g)
                    v.copy_(b1 * v + (1 - b1) * g * g)
                    step_size = lr * math.sqrt(1 - b1 ** n) / (1 - b0 ** n)
                    p.copy_(p - m * step_size / (torch.sqrt(v) + self.eps))
