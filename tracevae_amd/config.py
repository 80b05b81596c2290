"""Configuration tree of the train / test programs.

Field names, nesting and defaults follow the reference's `mltk.Config` classes so that
`train.sh` / `test.sh` flags (`--model.struct.z_dim=10` ...) and a saved `config.json`
mean the same thing here:

* `ExpConfig`               - `tracegnn/models/trace_vae/train.py:49-97`
* `TraceVAEConfig`          - `model/trace_vae.py:27-37`
* `TraceStructVAEConfig`    - `model/struct_vae.py:24-89`
* `TraceLatencyVAEConfig`   - `model/latency_vae.py:23-122`
* `GNNLayerConfig`          - `model/gnn_layers.py:25-33`
* `PoolingConfig`           - `model/pooling.py:25-31`
* `RealNVPFlowConfig`       - `model/realnvp_flow.py:10-15`

mltk itself is not available; this is a small re-implementation of the part that the two
scripts use: nested attribute classes, `--a.b.c=value` parsing, JSON round trip.
"""
import copy
import json
import os
from typing import Any, Dict, List, Optional

__all__ = [
    'Config', 'GNNLayerConfig', 'PoolingConfig', 'RealNVPFlowConfig',
    'TraceStructVAEConfig', 'TraceLatencyVAEConfig', 'TraceVAEConfig', 'ExpConfig',
    'parse_flags', 'config_to_dict', 'config_from_dict',
]


class Config(object):
    """Nested attribute container.  Class attributes are the defaults; nested `Config`
    subclasses (declared as inner classes or as instances) become per-instance children."""

    def __init__(self, **kwargs):
        for name, value in _declared_fields(type(self)).items():
            if isinstance(value, type) and issubclass(value, Config):
                value = value()
            else:
                value = copy.deepcopy(value)
            object.__setattr__(self, name, value)
        for k, v in kwargs.items():
            if k not in self.__dict__:
                raise AttributeError(f'{type(self).__name__} has no field {k!r}')
            setattr(self, k, v)

    def __setattr__(self, key, value):
        if key not in self.__dict__:
            raise AttributeError(f'{type(self).__name__} has no field {key!r}')
        object.__setattr__(self, key, value)

    def __repr__(self):
        return f'{type(self).__name__}({config_to_dict(self)!r})'

    def __eq__(self, other):
        return isinstance(other, Config) and config_to_dict(self) == config_to_dict(other)


def _declared_fields(cls) -> Dict[str, Any]:
    fields: Dict[str, Any] = {}
    for klass in reversed(cls.__mro__):
        if klass in (object, Config):
            continue
        for name, value in vars(klass).items():
            if name.startswith('_') or callable(value) and not (
                    isinstance(value, type) and issubclass(value, Config)):
                continue
            if isinstance(value, (staticmethod, classmethod, property)):
                continue
            fields[name] = value
    return fields


class GNNLayerConfig(Config):
    type = 'GAT'                 # 'GAT' | 'GraphConv'
    use_batch_norm = True

    class gat(Config):
        num_attention_heads = 2


class PoolingConfig(Config):
    use_batch_norm = True

    class attention(Config):
        hidden_layers: List[int] = []


class RealNVPFlowConfig(Config):
    flow_levels = 5
    coupling_hidden_layer_count = 1
    coupling_hidden_layer_units = 64
    coupling_layer_scale = 'sigmoid'
    strict_invertible = False


class TraceStructVAEConfig(Config):
    z_dim = 3
    realnvp = RealNVPFlowConfig
    use_batch_norm = True

    class encoder(Config):
        gnn = GNNLayerConfig
        gnn_layers: List[int] = [500, 500, 500, 500]
        pool_type = 'avg'        # 'root' | 'avg' | 'attention'
        pool_config = PoolingConfig
        z_logstd_min: Optional[float] = -7
        z_logstd_max: Optional[float] = 2
        use_posterior_flow = False

    class decoder(Config):
        use_prior_flow = False
        z_as_context = False
        use_depth = False
        use_idx = True
        gnn = GNNLayerConfig
        gnn_layers: List[int] = [500, 500, 500, 500]
        node_count_layers: List[int] = [500]
        graph_embedding_layers: List[int] = [500, 500]
        latent_embedding_size = 40


class TraceLatencyVAEConfig(Config):
    use_operation_embedding = True
    z2_dim = 10
    realnvp = RealNVPFlowConfig
    use_batch_norm = True

    class encoder(Config):
        gnn = GNNLayerConfig
        gnn_layers: List[int] = [500, 500, 500, 500]
        operation_embedding_stop_grad = True
        pool_type = 'avg'
        pool_config = PoolingConfig
        z2_logstd_min: Optional[float] = -7
        z2_logstd_max: Optional[float] = 2
        use_posterior_flow = False

    class decoder(Config):
        use_prior_flow = False
        z2_prior_mixtures = 1
        condition_on_z = True
        z2_given_z_stop_grad = True
        z2_given_z_hidden_layers: List[int] = [250, 250]
        z2_logstd_min: Optional[float] = -5
        z2_logstd_max: Optional[float] = 2
        gnn = GNNLayerConfig
        gnn_layers: List[int] = [500, 500, 500, 500]
        graph_embedding_layers: List[int] = [500, 500]
        latent_embedding_size = 40
        operation_embedding_stop_grad = True
        latency_logstd_min: Optional[float] = -7
        use_latency_mask = True
        clip_latency_to_one_dim = False
        use_biased_latency = False
        use_anomaly_detection_normal = False
        biased_normal_std_threshold = 4.0
        safe_normal_std_threshold = 6.0


class TraceVAEConfig(Config):
    operation_embedding_dim = 40
    arch = 'default'
    struct = TraceStructVAEConfig
    latency = TraceLatencyVAEConfig
    use_latency = True


class ExpConfig(Config):
    model = TraceVAEConfig
    device: Optional[str] = 'cpu'
    seed: Optional[int] = 0

    class train(Config):
        max_epoch = 60
        struct_pretrain_epochs: Optional[int] = 40
        ckpt_epoch_freq: Optional[int] = 5
        test_epoch_freq: Optional[int] = 5
        latency_hist_epoch_freq: Optional[int] = 10
        latency_std_hist_epoch_freq: Optional[int] = 5
        use_early_stopping = False
        val_epoch_freq: Optional[int] = 2
        kl_beta = 1.0
        warm_up_epochs: Optional[int] = None
        l2_reg = 0.0001
        z_unit_ball_reg: Optional[float] = None
        z2_unit_ball_reg: Optional[float] = None
        init_batch_size = 64
        batch_size = 64
        val_batch_size = 64
        optimizer = 'rmsprop'    # 'adam' | 'rmsprop'
        initial_lr = 0.001
        lr_anneal_ratio = 0.1
        lr_anneal_epochs = 30
        clip_norm: Optional[float] = None
        global_clip_norm: Optional[float] = 10
        test_n_z = 10
        num_plot_samples = 20

    class test(Config):
        batch_size = 64
        eval_n_z = 10
        use_biased = True
        latency_log_prob_weight = True
        clip_nll: Optional[float] = 100_000

    class report(Config):
        html_ext = '.html.gz'

    class dataset(Config):
        root_dir = os.path.abspath('./data/processed')


# ---------------------------------------------------------------------------------------
# flag parsing / (de)serialisation
# ---------------------------------------------------------------------------------------
def _coerce(text: str, default: Any) -> Any:
    low = text.strip().lower()
    if low in ('none', 'null', ''):
        return None
    if isinstance(default, bool):
        if low in ('true', '1', 'yes', 'on'):
            return True
        if low in ('false', '0', 'no', 'off'):
            return False
        raise ValueError(f'not a boolean: {text!r}')
    if isinstance(default, int) and not isinstance(default, bool):
        try:
            return int(text)
        except ValueError:
            return float(text)
    if isinstance(default, float):
        return float(text)
    if isinstance(default, (list, tuple)):
        value = json.loads(text) if text.strip().startswith('[') else \
            [json.loads(t) for t in text.split(',') if t.strip()]
        return list(value)
    if default is None:
        try:
            return json.loads(text)
        except ValueError:
            return text
    return text


def set_by_path(config: Config, path: str, text: str) -> None:
    node = config
    parts = path.split('.')
    for p in parts[:-1]:
        node = getattr(node, p)
        if not isinstance(node, Config):
            raise AttributeError(f'{path!r}: {p!r} is not a config section')
    leaf = parts[-1]
    if leaf not in node.__dict__:
        raise AttributeError(f'unknown config field: {path!r}')
    setattr(node, leaf, _coerce(text, getattr(type(node), leaf, getattr(node, leaf))))


def parse_flags(config: Config, argv: List[str], strict: bool = True) -> List[str]:
    """Apply `--a.b.c=value` (or `--a.b.c value`) flags to `config`; returns what was not
    consumed.  The reference accepts exactly this form (`train.sh:4`, `test_utils.py:36-66`)."""
    rest: List[str] = []
    i = 0
    while i < len(argv):
        arg = argv[i]
        if arg.startswith('--'):
            if '=' in arg:
                key, value = arg[2:].split('=', 1)
            elif i + 1 < len(argv) and not argv[i + 1].startswith('--'):
                key, value = arg[2:], argv[i + 1]
                i += 1
            else:
                key, value = arg[2:], 'true'
            key = key.replace('-', '_')
            try:
                set_by_path(config, key, value)
            except AttributeError:
                if strict:
                    raise
                rest.append(arg)
        else:
            rest.append(arg)
        i += 1
    return rest


def config_to_dict(config: Config) -> Dict[str, Any]:
    out = {}
    for k, v in config.__dict__.items():
        out[k] = config_to_dict(v) if isinstance(v, Config) else copy.deepcopy(v)
    return out


def config_from_dict(config: Config, values: Dict[str, Any], strict: bool = False) -> Config:
    for k, v in values.items():
        if k not in config.__dict__:
            if strict:
                raise AttributeError(f'unknown config field {k!r}')
            continue
        cur = getattr(config, k)
        if isinstance(cur, Config):
            if isinstance(v, dict):
                config_from_dict(cur, v, strict)
        else:
            setattr(config, k, v)
    return config
