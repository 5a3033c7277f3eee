python -m pytest tests/test_tile_losses.py -x -q -m gpu 2>&1 | grep -E "Mismatch|Max abs|Max rel|passed|failed|Error|error" | head -20
python tools/diag_tile.py cos 512
NMSA_TILE_ABLATE=2 python tools/diag_tile.py cos 512
NMSA_TILE_ABLATE=10 python tools/diag_tile.py cos 512
python tools/diag_tile.py cos 256
python tools/diag_tile.py cos 768
python tools/diag_tile.py ce 150
python tools/diag_tile.py ce 64
