"""ModelParams with the reference's INI surface (misc/utils.py:11-73): `model`, `coordinates`
in {polar, cartesian}, `quantization_step`; the quantiser objects are this package's on-device ones."""
from __future__ import annotations

import configparser

from .quantization import CartesianQuantizer, PolarQuantizer


class ModelParams:
    def __init__(self, model_params_path=None, *, model: str = "egonn", coordinates: str = "polar",
                 quantization_step=None, **minkloc_kwargs):
        if model_params_path is not None:
            config = configparser.ConfigParser()
            read = config.read(model_params_path)
            if not read:
                raise FileNotFoundError(model_params_path)
            params = config['MODEL']
            self.model_params_path = model_params_path
            self.model = params.get('model')
            self.output_dim = params.getint('output_dim', 256)
            self.coordinates = params.get('coordinates', 'polar')
            raw_step = params.get('quantization_step', None)
            mk = {k: params[k] for k in ('feature_size', 'planes', 'layers', 'num_top_down', 'conv0_kernel_size',
                                         'block', 'pooling') if k in params}
        else:
            mk = dict(minkloc_kwargs)
            self.model_params_path = None
            self.model = model
            self.output_dim = 256
            self.coordinates = coordinates
            raw_step = quantization_step
        assert self.coordinates in ['polar', 'cartesian'], f'Unsupported coordinates: {self.coordinates}'
        if 'polar' in self.coordinates:
            if isinstance(raw_step, str):
                raw_step = [float(e) for e in raw_step.split(',')]
            self.quantization_step = [float(e) for e in raw_step]
            assert len(self.quantization_step) == 3, \
                'Expected 3 quantization steps: for sectors (degrees), rings (meters) and z coordinate (meters)'
            self.quantizer = PolarQuantizer(quant_step=self.quantization_step)
        else:
            self.quantization_step = float(raw_step)
            self.quantizer = CartesianQuantizer(quant_step=self.quantization_step)
        if 'MinkLoc' in self.model:                       # reference misc/utils.py:41-58
            def ints(v, default):
                if v is None:
                    return default
                return [int(e) for e in v.split(',')] if isinstance(v, str) else [int(e) for e in v]
            self.feature_size = int(mk.get('feature_size', 256))
            self.planes = ints(mk.get('planes'), [32, 64, 64])
            self.layers = ints(mk.get('layers'), [1, 1, 1])
            self.num_top_down = int(mk.get('num_top_down', 1))
            self.conv0_kernel_size = int(mk.get('conv0_kernel_size', 5))
            self.block = mk.get('block', 'BasicBlock')
            self.pooling = mk.get('pooling', 'GeM')

    def print(self):
        print('Model parameters:')
        for k, v in vars(self).items():
            if k == 'quantization_step':
                if self.coordinates == 'polar':
                    print(f'quantization_step - sector: {v[0]} [deg] / ring: {v[1]} [m] / z: {v[2]} [m]')
                else:
                    print(f'quantization_step: {v} [m]')
            elif k != 'quantizer':
                print(f'{k}: {v}')
        print('')
