"""hk.nets stand-in: MLP (see haiku/__init__.py)."""


class _Lazy:
  pass


def __getattr__(name):
  if name == "MLP":
    import haiku as hk

    class MLP(hk.Module):
      def __init__(self, output_sizes, w_init=None, b_init=None, with_bias=True, activation=None,
                   activate_final=False, name=None):
        super().__init__(name=name)
        self.activation = activation
        self.activate_final = activate_final
        self.layers = [hk.Linear(n, with_bias=with_bias, name=f"linear_{i}")
                       for i, n in enumerate(output_sizes)]

      def __call__(self, x):
        n = len(self.layers)
        for i, layer in enumerate(self.layers):
          x = layer(x)
          if i < n - 1 or self.activate_final:
            x = self.activation(x)
        return x

    globals()["MLP"] = MLP
    return MLP
  raise AttributeError(name)
