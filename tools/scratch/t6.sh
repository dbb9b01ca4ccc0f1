set -e
A="MSPI_DW_WLDS=0" B="X=1" bash tools/scratch/ab.sh
A="MSPI_DW_TILE_ALL=1" B="X=1" bash tools/scratch/ab.sh
