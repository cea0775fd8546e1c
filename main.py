#!/usr/bin/env python
"""`python main.py --model resnet --model-config "{'depth': 50}" ...` - same entry point as the
reference's main.py; the implementation lives in convnet.pytorch_amd/main.py."""
import convnet_amd  # noqa: F401  (registers the package living in ./convnet.pytorch_amd)
from convnet_amd.main import main

if __name__ == '__main__':
    main()
