from .default_config import CfgNode, get_cfg_defaults, get_lamp_config, get_model_defaults, load_yaml_into  # noqa: F401
