"""absl stand-in (logging only)."""
import logging as _logging


class _Logging:
  info = staticmethod(_logging.getLogger("absl").info)
  warning = staticmethod(_logging.getLogger("absl").warning)
  error = staticmethod(_logging.getLogger("absl").error)
  debug = staticmethod(_logging.getLogger("absl").debug)

  @staticmethod
  def log_first_n(level, msg, n, *args):
    pass

  @staticmethod
  def flush():
    pass

  INFO, WARNING, ERROR = 0, 1, 2


logging = _Logging()
