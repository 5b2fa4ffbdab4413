from newtonnet_amd.data.ingest import Frame, MolecularStatistics, collate, read_extxyz
