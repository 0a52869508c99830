"""ModelParams with the reference's INI surface (misc/utils.py:11-73): `model`, `coordinates`
in {polar, cartesian}, `quantization_step`; the quantiser objects are this package's on-device ones."""
from __future__ import annotations

import configparser

from .quantization import CartesianQuantizer, PolarQuantizer


class ModelParams:
    def __init__(self, model_params_path=None, *, model: str = "egonn", coordinates: str = "polar",
                 quantization_step=None):
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
        else:
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
