from . import feeder_nucla_gcn   # noqa: F401
