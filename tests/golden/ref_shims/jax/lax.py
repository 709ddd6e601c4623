"""jax.lax stand-in: only names referenced in annotations of imported-but-unexecuted code."""


class Precision:  # noqa: D101
  DEFAULT = HIGH = HIGHEST = None


def stop_gradient(x):
  return x
