"""Host mirror of audio_models/ConvNets_SpeechCommands/create_model.py (ref l.8-17).

create_model(path): a path containing 'ConvNets_SpeechCommands' holds a pickled DataParallel whose
`.module` is the classifier (adv_train_speech_commands.py:108,336-340); any other path holds a bare
pickled module (the bundled M5 checkpoints).  Returns the module in float32 / eval mode.  Like the
reference it makes the model-definition packages importable for unpickling (`models.vgg`, `M5Net`);
unlike the reference it resolves them next to this file instead of through the working directory,
and it unpickles with an allow-list (torch >= 2.6 refuses arbitrary globals by default)."""
import collections
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(os.path.dirname(_HERE), 'M5'), _HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _allowed_globals():
    """Everything a pickled M5 / DataParallel(VGG) / DataParallel(CifarResNeXt) module references: the model classes of
    this package (same module paths as the reference's: `M5Net`, `models.vgg`, `models.resnext`) and the torch.nn layers
    they are built from.  certified_robustness_eval.py:57-59 loads the ResNeXt29 checkpoint by default."""
    import M5Net
    from models import resnext, vgg
    nn = torch.nn
    return [M5Net.M5, vgg.VGG, resnext.CifarResNeXt, resnext.ResNeXtBottleneck, nn.DataParallel, nn.Sequential, nn.ModuleList,
            nn.Conv1d, nn.Conv2d, nn.BatchNorm1d, nn.BatchNorm2d, nn.MaxPool1d, nn.MaxPool2d, nn.AvgPool2d, nn.AdaptiveAvgPool2d,
            nn.ReLU, nn.Dropout, nn.Linear, nn.Identity, set, collections.OrderedDict, torch.device]


def create_model(path):
    with torch.serialization.safe_globals(_allowed_globals()):
        obj = torch.load(path, map_location='cpu', weights_only=True)
    model = obj.module if 'ConvNets_SpeechCommands' in path else obj
    model.float()
    model.eval()
    return model
