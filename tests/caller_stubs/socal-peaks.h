/* reference standalone.c:494-495 includes a list of peaks that its query-peaks-from-osm.py generates (network): none here */
