import sys

from neighborhoodwatch_amd.neighborhoodwatch import main

sys.exit(main())
