import torch

torch_dtypes = {'float': torch.float, 'float32': torch.float32, 'float64': torch.float64,
                'double': torch.double, 'float16': torch.float16, 'half': torch.half}
