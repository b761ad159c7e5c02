from neighborhoodwatch_amd.neighborhoodwatch import main
main()
