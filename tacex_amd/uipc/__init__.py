from .uipc_sim import UipcSim, UipcSimCfg
from .uipc_object import UipcObject, UipcObjectCfg
from .uipc_attachments import UipcIsaacAttachments, UipcIsaacAttachmentsCfg

__all__ = ["UipcSim", "UipcSimCfg", "UipcObject", "UipcObjectCfg", "UipcIsaacAttachments", "UipcIsaacAttachmentsCfg"]
